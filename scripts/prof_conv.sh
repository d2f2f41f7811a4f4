#!/bin/bash
# GPU-box tool: PMC counters for selected conv layers (separate passes, kernel-trace only)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
rocprofv3 -L > $R/gpurun_out/prof/counters.txt 2>&1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
            "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/prof/pmc_$tag -- python3 $R/scripts/tune_conv.py --reps 2 --tiles ${TILES:-64x128} --only "${ONLY:-conv2/3x3/}" > $R/gpurun_out/prof/pmc_$tag.log 2>&1
  echo "pass $tag rc=$?"
done
find $R/gpurun_out/prof -name "*counter_collection.csv" | head
