#!/bin/bash
# GPU-box routine: rocprofv3 kernel-trace statistics of one forward pass at a time on ONE stream (a launch's duration is its own), top kernels printed.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-extra --min-seconds 0 --requests 1 --streams 1 > $OUT/stats.log 2>&1
echo "rc=$?"
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:28]:
    print('{:70s} calls {:5s} avg {:9.2f} us  total {:8.3f} ms  {:5.2f} %'.format(r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
PY
