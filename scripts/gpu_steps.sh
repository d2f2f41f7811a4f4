#!/bin/bash
# GPU-box routine: run the steps listed in a file (one per line: "<seconds> <logfile> <command ...>"), each under its own timeout,
# logs under gpurun_out/.  A step that fails with an ordinary status (a failed assertion) does not stop the list; a step that is
# killed by its timeout or by a signal (rc >= 124: a hang, a GPU fault, an abort) does -- never start another GPU step after that.
set -u
mkdir -p gpurun_out
: > gpurun_out/steps.log
worst=0
while IFS= read -r line; do
    [ -z "$line" ] && continue
    case "$line" in \#*) continue;; esac
    secs=${line%% *}; rest=${line#* }; log=${rest%% *}; cmd=${rest#* }
    echo "=== [$secs s] $cmd > gpurun_out/$log" | tee -a gpurun_out/steps.log
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$log" 2>&1
    rc=$?
    echo "=== rc=$rc" | tee -a gpurun_out/steps.log
    tail -n 6 "gpurun_out/$log"
    [ $rc -gt $worst ] && worst=$rc
    if [ $rc -ge 124 ]; then echo "STOPPING after rc=$rc (timeout / signal)" | tee -a gpurun_out/steps.log; exit $rc; fi
done < "$1"
exit $worst
