#!/bin/bash
# GPU-box tool: PMC counters of the pointwise kernel and the general kernel on one layer shape (separate passes, kernel-trace only)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_pw
rm -rf $O; mkdir -p $O
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
            "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
            "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES_EQ_64" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TD_BUSY_avr"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$tag -- python3 $R/scripts/time_pw.py --rounds 1 --reps 2 --tiles 0 --only "${ONLY:-3b siblings}" > $O/pmc_$tag.log 2>&1
  echo "pass $tag rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'conv_' not in k or 'pack' in k: continue
        agg[k[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print('   {:32s} n={:3d} mean={:.4g}'.format(c, len(v), sum(v) / len(v)))
PY
