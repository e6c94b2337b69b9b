#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "== GPU suite"
SECONDS=0
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r05_gputests_6.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"; tail -22 gpurun_out/r05_gputests_6.log | cut -c1-200
echo "== profile round"
bash tools/profile_round.sh r05 cant > gpurun_out/r05_profile_round.log 2>&1; echo "profile_round rc=$?"; tail -5 gpurun_out/r05_profile_round.log | cut -c1-400
cat gpurun_out/prof_r05_readme_line.txt
echo "== driver form"
bash tools/driver_form_runs.sh r05a 3
