#!/bin/bash
# counter passes over scripts/run_wino_once.py (separate --pmc passes with --kernel-trace only): L2 and L1 hit / miss / stall counters of
# conv_wino4_kernel.  (A fifth pass with TA_* counters hung the profiler on this pool: left out.)  Do not pipe into tail: gpurun kills silent runs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_wino
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/scripts/run_wino_once.py 4 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $set" | tee -a $OUT/progress.log
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_wino4' not in r['Kernel_Name']: continue
        key = 'grid%s' % r.get('Grid_Size', '?')
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for key in agg:
    print(key)
    for c, v in sorted(agg[key].items()):
        print('   {:40s} {:.4g} per launch'.format(c, v / max(1, cnt[(key, c)])))
PY
