# usage (GPU box): bash tools/c1_ab.sh — the C1-size step (BASELINE configs[0]) with the 128^2 GEMM's loop forced to each form, alternating in one job:
# SVI_GEMM_PF = 1 (one K tile of loads in flight, rounds 1-6), 4 (four, four waves), 0 (default: four on eight waves when the launch is at most one workgroup per CU)
set -u
mkdir -p gpurun_out
for pf in 1 4 0 1 4 0; do
  SVI_GEMM_PF=$pf timeout 300 python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline --no-vendor --no-full-clip --no-vae 2>/dev/null | grep '"metric"' | tail -1 > gpurun_out/c1_ab_pf$pf.json
  python - $pf <<'PY'
import json, sys
j = json.load(open(f"gpurun_out/c1_ab_pf{sys.argv[1]}.json"))
print("SVI_GEMM_PF", sys.argv[1], "ms_per_step", j["ms_per_step"], {k: v.get("ms_per_step") for k, v in j["roofline_all"].items()})
PY
done
