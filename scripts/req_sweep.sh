#!/bin/bash
# GPU-box routine: the bench command with 4 .. 8 whole-batch requests in flight on ONE box (the number is a property of the kernel mix: re-sweep when a
# kernel's occupancy changes) -> gpurun_out/req_sweep.txt
for r in 4 5 6 7 8; do
  line=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-images 0 --no-extra --requests $r 2>/dev/null | grep '^{' | tail -1)
  echo "requests=$r $(echo "$line" | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print(round(d["value"], 1), "images/s", round(d["ms_per_step"], 4), "ms/step")')" | tee -a gpurun_out/req_sweep.txt
done
