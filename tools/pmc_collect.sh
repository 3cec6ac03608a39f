#!/bin/bash
# usage: tools/pmc_collect.sh <probe> <outdir>   — separate rocprofv3 --pmc passes (never combined with tracing)
set -u
P=$1; OUT=$2; R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS -d $R/$OUT/pass$i -o p --output-format csv -- python $R/tools/kernel_probe.py $P 2 > $R/$OUT/pass$i.log 2>&1
  echo "pass $i ($CTRS): rc=$?"
done
cd $R
find $OUT -name "*counter_collection.csv" | head
