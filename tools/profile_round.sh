#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default-shaped bench (2 steps) -> gpurun_out/<tag>_kernel_stats.md
#   2. separate rocprofv3 --pmc passes of the dominant kernel alone (tools/kernel_probe.py attn)
#      -> gpurun_out/<tag>_flash_pmc.txt and gpurun_out/<tag>_flash_pmc.json (per-launch HBM bytes, corrected as
#      MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE is in KiB and reads half of a wide coalesced stream on gfx950 -> x2)
#   3. the same passes for the ffn1 GEMM -> gpurun_out/<tag>_gemm_ffn1_pmc.txt;  4. source hashes -> gpurun_out/<tag>_source_hashes.json
# --pmc is never combined with tracing options.
set -u
TAG=${1:-r1}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- $CMD > $R/gpurun_out/prof_$TAG/run.log 2>&1
DB=$(find $R/gpurun_out/prof_$TAG -name "*_results.db" | head -1)
python $R/tools/rocpd_summary.py "$DB" $R/gpurun_out/${TAG}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae" > /dev/null
grep '"metric"' $R/gpurun_out/prof_$TAG/run.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json
rm -f "$DB"        # the database is large; the summary is what gets committed
cd $R
tools/pmc_collect.sh attn gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$TAG flash > gpurun_out/${TAG}_flash_pmc.txt
python - <<PY
import hashlib, json, re
# one self-attention call = the optimistic pass + the flagged second pass (two instantiations of flash_fwd2_kernel): per call the
# counters of both are added (the second pass exits at once on benign operands and contributes next to nothing)
vals, kernels, cur = {}, [], None
for line in open("gpurun_out/${TAG}_flash_pmc.txt"):
    if line and not line[0].isspace():
        cur = line.strip(); kernels.append(cur)
        continue
    m = re.match(r"\s+(\S+)\s+per-dispatch\s+([0-9.]+)", line)
    if m and cur and "flash_fwd2_kernel" in cur:
        vals[m.group(1)] = vals.get(m.group(1), 0.0) + float(m.group(2))
fetch_kib, write_kib = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
out = {"kernel": "flash_fwd2_kernel (self-attention, L=32760, 12 heads; optimistic pass + flagged second pass)", "kernels_summed": [k for k in kernels if "flash_fwd2_kernel" in k], "counters_per_launch": vals,
       "attention_src_sha": hashlib.sha256(open("stable-video-infinity_amd/csrc/svi_attention.hip", "rb").read()).hexdigest()[:16],
       "fetch_bytes": None if fetch_kib is None else fetch_kib * 1024 * 2, "write_bytes": None if write_kib is None else write_kib * 1024,
       "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section: wide coalesced reads are tallied at half); separate --pmc passes, tools/pmc_collect.sh"}
if out["fetch_bytes"] is not None and out["write_bytes"] is not None:
    out["hbm_bytes"] = out["fetch_bytes"] + out["write_bytes"]
json.dump(out, open("gpurun_out/${TAG}_flash_pmc.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
rm -rf gpurun_out/pmc_$TAG/pass*/  # raw CSVs are large; the summary is kept
# 3. the same PMC passes for the largest GEMM (ffn1: M = 32760, N = 8960, K = 1536, GELU epilogue) -> gpurun_out/<tag>_gemm_ffn1_pmc.txt
tools/pmc_collect.sh gemm_ffn1 gpurun_out/pmcg_$TAG > gpurun_out/pmcg_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcg_$TAG gemm > gpurun_out/${TAG}_gemm_ffn1_pmc.txt
rm -rf gpurun_out/pmcg_$TAG/pass*/
# 4. which sources these summaries were collected on
python - <<PY
import hashlib, json
srcs = ["svi_attention.hip", "svi_gemm.hip", "svi_dit.hip", "svi_elementwise.hip", "svi_vae.hip"]
json.dump({s: hashlib.sha256(open("stable-video-infinity_amd/csrc/" + s, "rb").read()).hexdigest()[:16] for s in srcs}, open("gpurun_out/${TAG}_source_hashes.json", "w"), indent=1)
PY
