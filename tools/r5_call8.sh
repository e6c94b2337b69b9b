#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "== overlap probe"
timeout -k 10 200 python3 tools/overlap_probe.py cant 500 1 2 3 4 2>&1 | tail -1
timeout -k 10 200 python3 tools/overlap_probe.py G3_circuit 200 1 2 2>&1 | tail -1
echo "== GPU suite (driver form: -x -q)"
SECONDS=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r05_gputests_8.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"; tail -12 gpurun_out/r05_gputests_8.log | cut -c1-200
