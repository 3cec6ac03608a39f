#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/final_lines.sh <tag>
# The round's closing artefacts on the frozen tree, beside tools/profile_round.sh <tag>: the full -m gpu suite, smoke(), and the bench lines
# (driver-shaped default, eager, C1, C4, and the opt-in fp8 lines) -> gpurun_out/<tag>_*; copy them to profiles/ afterwards.
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -z "${LINES_ONLY:-}" ]; then      # LINES_ONLY=1: only the bench lines (after a bench.py-only change on an already tested tree)
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests.log 2>&1
tail -3 gpurun_out/${TAG}_gputests.log
cp gpurun_out/parity_report.jsonl gpurun_out/${TAG}_parity_report.jsonl 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/${TAG}_smoke.log 2>&1
tail -1 gpurun_out/${TAG}_smoke.log
fi
line() { # name, args...
  local name=$1; shift
  timeout 900 python bench.py "$@" 2> gpurun_out/${TAG}_bench_${name}.err | grep '"metric"' | tail -1 > gpurun_out/${TAG}_bench_${name}.json
  python - "$name" gpurun_out/${TAG}_bench_${name}.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2]))
    c = j["config"]
    print(sys.argv[1], j["ms_per_step"], j["value"], (j.get("roofline") or {}).get("frac"), (j.get("roofline") or {}).get("traffic"),
          {k: c.get(k) for k in ("full_clip_s", "value_full_clip", "full_clip_steady_s") if c.get(k) is not None}, (c.get("window") or {}).get("per_clip_fixed_cost_s"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
line default --gpus 1 --steps 20 --warmup 5
# the metric as defined, as a workload: BASELINE configs[2] on one GPU (3 clips, seeds k x 42, 2 prompts cycled, decode + 8-bit frames + stitch), resident loop
# (one captured step graph for the window) against a loop that re-captures per clip; then the same window sharded over 2 ranks that share the GPU (gloo probe)
line window --steps 5 --warmup 2 --window 3 --window-ab --no-cpu-baseline --no-full-clip
line window_gloo_x2 --gpus 2 --transport gloo --steps 3 --warmup 1 --window 2 --no-cpu-baseline --no-full-clip
line c1_window --workload c1 --steps 10 --warmup 3 --window 6 --window-ab --no-cpu-baseline --no-full-clip
SVI_CROSS_FUSED=0 line cross_two_kernels --steps 10 --warmup 3 --no-cpu-baseline --no-vendor
line no_graph --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-vendor
line fp8_attn --steps 10 --warmup 3 --fp8-attn --no-cpu-baseline --no-vendor
line fp8_mfma --steps 10 --warmup 3 --fp8-mfma --no-cpu-baseline --no-vendor
line fp8_mfma_attn --steps 10 --warmup 3 --fp8-mfma --fp8-attn --no-cpu-baseline --no-vendor
line fp8_all --steps 10 --warmup 3 --fp8-all --no-cpu-baseline --no-vendor
line c1 --workload c1 --steps 10 --warmup 3 --no-cpu-baseline
line c4 --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-vendor
# round 6: the non-benign lines (peaky attention logits through learned-gain factors, a 200-token prompt on the streaming cross-attention kernel), the step beside a
# live one-rank RCCL communicator, and the kernel-level sweeps they summarise
line attn_gain_2 --steps 6 --warmup 2 --no-cpu-baseline --no-full-clip --no-vendor --no-vae --attn-gain 2
line attn_gain_3 --steps 6 --warmup 2 --no-cpu-baseline --no-full-clip --no-vendor --no-vae --attn-gain 3
line prompt_tokens_200 --steps 6 --warmup 2 --no-cpu-baseline --no-full-clip --no-vendor --no-vae --prompt-tokens 200
line rccl_probe --steps 6 --warmup 2 --no-cpu-baseline --no-full-clip --no-vendor --no-vae --rccl-probe
timeout 600 python tools/attn_stats.py --json gpurun_out/${TAG}_attn_stats.json > gpurun_out/${TAG}_attn_stats.txt 2>&1
timeout 400 python tools/yardstick.py 7 --json gpurun_out/${TAG}_yardstick.json > gpurun_out/${TAG}_yardstick.txt 2>&1
tail -8 gpurun_out/${TAG}_yardstick.txt
