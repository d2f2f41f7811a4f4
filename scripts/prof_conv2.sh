#!/bin/bash
# GPU-box tool: TA / TCP / SQ counters for one conv layer and a list of kernel variants (tune_conv.py tile names)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof2; mkdir -p $R/gpurun_out/prof2
i=0
for pass in "TA_BUSY_avr GRBM_GUI_ACTIVE" \
            "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum" \
            "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/prof2/p$i -- python3 $R/scripts/tune_conv.py --reps 2 --tiles ${TILES} --only "${ONLY}" > $R/gpurun_out/prof2/p$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc" | tee -a $R/gpurun_out/prof2/passes.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout - stop"; exit 99; fi
done
