set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
for pf in 1 0 1 0; do
  SVI_GEMM_PF=$pf timeout 300 python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline --no-vendor --no-full-clip --no-vae 2>/dev/null | grep '"metric"' | tail -1 > gpurun_out/r6x_c1_pf$pf.json
  python - $pf <<'PY'
import json, sys
j = json.load(open(f"gpurun_out/r6x_c1_pf{sys.argv[1]}.json"))
print("pf", sys.argv[1], j["ms_per_step"], {k: v.get("ms_per_step") for k, v in j["roofline_all"].items()})
PY
done
