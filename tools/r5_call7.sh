#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "== GPU suite (driver form: -x -q)"
SECONDS=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r05_gputests_7.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"; tail -18 gpurun_out/r05_gputests_7.log | cut -c1-200
echo "== smoke"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== driver form"
bash tools/driver_form_runs.sh r05b 3
