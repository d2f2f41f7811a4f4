#!/bin/bash
# GPU-box routine: the bench command with an environment variable at two values, alternating on ONE box (boxes of the pool differ by ~3 %).
# usage: scripts/ab_bench.sh VAR A B [pairs]     -> gpurun_out/ab_<VAR>.txt (one line per run: value, images/s, ms per step)
set -u
VAR=$1; A=$2; B=$3; PAIRS=${4:-2}
mkdir -p gpurun_out
OUT=gpurun_out/ab_$VAR.txt
: > $OUT
for i in $(seq 1 $PAIRS); do
  for v in $A $B; do
    line=$(env $VAR=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-images 0 --no-extra 2>/dev/null | grep '^{' | tail -1)
    rc=$?
    echo "$VAR=$v $(echo "$line" | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print(round(d["value"], 1), "images/s", round(d["ms_per_step"], 4), "ms/step", "single", round(d.get("single_request_ms_per_infer", 0), 4), "conv", d["per_op"]["Convolution"]["ms_per_step"], "frac", round(d["roofline"]["frac"], 4))' 2>/dev/null || echo "failed rc=$rc")" | tee -a $OUT
  done
done
