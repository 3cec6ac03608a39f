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
rm -rf gpurun_out/pmc_$TAG/pass*/  # raw CSVs are large; the summary is kept
# 3. the same PMC passes for the largest GEMM (ffn1: M = 32760, N = 8960, K = 1536, GELU epilogue) -> gpurun_out/<tag>_gemm_ffn1_pmc.txt
tools/pmc_collect.sh gemm_ffn1 gpurun_out/pmcg_$TAG > gpurun_out/pmcg_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcg_$TAG gemm > gpurun_out/${TAG}_gemm_ffn1_pmc.txt
rm -rf gpurun_out/pmcg_$TAG/pass*/
# 3b. and for ffn2 (M = 32760, N = 1536, K = 8960, gate + residual epilogue) -> gpurun_out/<tag>_gemm_ffn2_pmc.txt
tools/pmc_collect.sh gemm_ffn2 gpurun_out/pmch_$TAG > gpurun_out/pmch_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmch_$TAG gemm > gpurun_out/${TAG}_gemm_ffn2_pmc.txt
rm -rf gpurun_out/pmch_$TAG/pass*/
# 3c. the opt-in fp8 QK^T attention: the attention probe under SVI_ATTN_QK8=1 -> gpurun_out/<tag>_flash_qk8_pmc.txt
SVI_ATTN_QK8=1 tools/pmc_collect.sh attn gpurun_out/pmcq_$TAG > gpurun_out/pmcq_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcq_$TAG "flash|mx8_quantize" > gpurun_out/${TAG}_flash_qk8_pmc.txt
rm -rf gpurun_out/pmcq_$TAG/pass*/
# 3d. the fused cross-attention (q RMS-normalised as it is read, K / V^T resident in LDS): HBM-bound -> gpurun_out/<tag>_flash_cross_pmc.txt
tools/pmc_collect.sh cross gpurun_out/pmcx_$TAG > gpurun_out/pmcx_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcx_$TAG "flash_cross|row_rs" > gpurun_out/${TAG}_flash_cross_pmc.txt
rm -rf gpurun_out/pmcx_$TAG/pass*/
# 3e. the VAE decoder's convolutions (3 latent frames at 480x832: every layer at its C2 spatial size) -> gpurun_out/<tag>_vae_conv_pmc.txt
tools/pmc_collect.sh vae gpurun_out/pmcv_$TAG > gpurun_out/pmcv_$TAG.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcv_$TAG "conv_dma2h" > gpurun_out/${TAG}_vae_conv_pmc.txt
rm -rf gpurun_out/pmcv_$TAG/pass*/
# 4. JSON summaries (per-launch HBM bytes, mfma_busy_in_clock, L2 hit rate) + the hashes of the sources they were collected on
python tools/pmc_to_json.py $TAG
