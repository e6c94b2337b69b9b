#!/bin/bash
# round 5, first GPU call: the lean merge kernel's forms A/B (cant-like headline shape first), then the GPU suite with durations
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python3 -c "import torch; print(torch.cuda.get_device_name(0))"
tools/ab_roll.sh cant 3 --steps 1000 --warmup 100 --windows 11
tools/ab_roll.sh cant3 2 --workload cant3 --steps 1000 --warmup 100 --windows 11
tools/ab_roll.sh cant_w512 1 --variant merge --wg 512 --items 8 --steps 1000 --warmup 100 --windows 11
tools/ab_roll.sh g3 1 --workload G3_circuit --steps 500 --warmup 50 --windows 11
tools/ab_roll.sh atm 1 --workload atmosmodd --steps 500 --warmup 50 --windows 11
tools/ab_roll.sh cant20 2 --steps 20 --warmup 5
echo "== GPU suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r05_gputests_1.log 2>&1; echo "pytest rc=$?"; tail -70 gpurun_out/r05_gputests_1.log
