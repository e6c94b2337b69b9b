#!/bin/bash
# The GPU suite exactly as the driver runs it, timed (run on the GPU box):  tools/suite_time.sh <tag>
# prints the wall time, the ten slowest tests and the seconds per test file.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
log=gpurun_out/gputests_${1:-x}.log
SECONDS=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=0 > $log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"
grep -E "^[0-9.]+s (call|setup|teardown)" $log | sort -rn | sed -n 1,10p | cut -c1-160
grep -E "^[0-9.]+s (call|setup|teardown)" $log | awk '{split($3,a,"::"); t[a[1]]+=substr($1,1,length($1)-1)} END {for (f in t) printf "%8.1fs %s\n", t[f], f}' | sort -rn
tail -1 $log
