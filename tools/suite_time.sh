#!/bin/bash
# The GPU suite exactly as the driver runs it, timed (run on the GPU box):  tools/suite_time.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
SECONDS=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/gputests_${1:-x}.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"; tail -14 gpurun_out/gputests_${1:-x}.log | cut -c1-160
