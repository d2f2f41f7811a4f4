#!/bin/bash
# GPU-box routine: the FP16 entry of bench.py (GoogLeNet as an FP16 IR, batch 256, eager passes on one stream: `bench.py --fp16-passes N`) under
# rocprofv3: kernel-trace statistics, then HBM traffic and SQ counters in their own passes (never combined with other trace domains).
# Raw output under gpurun_out/prof_fp16_<tag>; summaries: python scripts/summarize_fp16_profile.py gpurun_out/prof_fp16_<tag> <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r05}
OUT=$R/gpurun_out/prof_fp16_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --fp16-passes 8 > $OUT/stats.log 2>&1
echo "rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --fp16-passes 3 > $OUT/pmc_$c.log 2>&1
  echo "rc=$?"
done
echo "== pmc SQ pass"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_SQ -- python3 $R/bench.py --fp16-passes 3 > $OUT/pmc_SQ.log 2>&1
echo "rc=$?"
cd $R && python3 scripts/summarize_fp16_profile.py $OUT $TAG
