#!/bin/bash
# GPU-box routine: rocprofv3 kernel-trace statistics of the bench command, then HBM traffic counters in
# their own passes (never combined with other trace domains).  Writes raw output under gpurun_out/ and
# the judged summaries under profiles/ via scripts/summarize_profile.py.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r05}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# Two kernel-trace runs: the default bench command (8 whole-batch requests in flight: kernels of different passes
# overlap, so a launch's start-to-end time includes its neighbours) and the same workload one request at a time on
# ONE stream, where a launch's duration is its own -- that one is what bench.py's roofline (sampled steps that run
# alone on one stream) must agree with.
echo "== kernel trace + stats, default (requests in flight)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_forked -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-extra --min-seconds 0 > $OUT/stats_forked.log 2>&1
echo "rc=$?"; grep '^{' $OUT/stats_forked.log | tail -1 > $OUT/bench_line_forked.json
echo "== kernel trace + stats, one stream"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-extra --min-seconds 0 --requests 1 --streams 1 > $OUT/stats.log 2>&1
echo "rc=$?"; grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_line_under_profiler.json
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-extra --min-seconds 0 --no-node-timing --requests 1 --streams 1 > $OUT/pmc_$c.log 2>&1
  echo "rc=$?"
done
echo "== pmc SQ pass"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_SQ -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-extra --min-seconds 0 --no-node-timing --requests 1 --streams 1 > $OUT/pmc_SQ.log 2>&1
echo "rc=$?"
cd $R && python3 scripts/summarize_profile.py $OUT $TAG
