#!/bin/bash
# GPU-box routine: parity tests -> smoke -> bench.  Stops at the first step that fails -- killed by its timeout, by a
# signal (a GPU fault / abort: rc >= 128) or with any other non-zero status: never start another GPU step after a hang or
# a fault -- and exits with that step's status.
set -u
mkdir -p gpurun_out
step() {  # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "=== $* (limit ${secs}s)" | tee -a gpurun_out/ci.log
    timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
    local rc=$?
    echo "=== rc=$rc" | tee -a gpurun_out/ci.log
    tail -n 25 "gpurun_out/$log"
    if [ $rc -ne 0 ]; then echo "FAILED (rc=$rc): $* -- stopping; last lines of gpurun_out/$log above" | tee -a gpurun_out/ci.log; exit $rc; fi
}
: > gpurun_out/ci.log
step 900 test_gpu.log python -m pytest tests -m gpu -q -x --timeout=600 ${PYTEST_EXTRA:-}
step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 600 bench.log python bench.py --gpus 1 --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5}
exit 0
