#!/bin/bash
# GPU-box routine: rocprofv3 kernel-trace statistics of the bench command, then HBM traffic counters in
# their own passes (never combined with other trace domains).  Writes raw output under gpurun_out/ and
# the judged summaries under profiles/ via scripts/summarize_profile.py.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r01}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace + stats" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-images 0 > $OUT/stats.log 2>&1
echo "rc=$?"; grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_line_under_profiler.json
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-node-timing > $OUT/pmc_$c.log 2>&1
  echo "rc=$?"
done
echo "== pmc SQ pass"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-node-timing > $OUT/pmc_SQ.log 2>&1
echo "rc=$?"
cd $R && python3 scripts/summarize_profile.py $OUT $TAG
