#!/bin/bash
# GPU-box routine: HBM read traffic (rocprofv3 --pmc FETCH_SIZE with --kernel-trace only) of the 14- / 7-wide shared-V six-point launches with their
# tiles in patch-block-major (PVHIP_TUNE3=2) and channel-pair-major order (=1) -> gpurun_out/wino_order/{2,1}; summarise locally with
# python scripts/summarize_wino_order.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/wino_order
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for o in 2 1; do
  export PVHIP_TUNE3=$o
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$o -- python3 $R/scripts/run_wino_layers.py > $OUT/$o.log 2>&1
  echo "order $o rc=$?"
done
