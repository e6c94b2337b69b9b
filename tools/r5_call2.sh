#!/bin/bash
# round 5, second GPU call: rolling row sums with the products pinned to their step, the aliased window in the solver
# passes, the GPU suite with durations, the triangular solves next to MKL
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tools/ab_roll.sh cant_b 2 --steps 1000 --warmup 100 --windows 11
tools/ab_roll.sh cant3_b 1 --workload cant3 --steps 1000 --warmup 100 --windows 11
tools/ab_roll.sh g3_b 1 --workload G3_circuit --steps 500 --warmup 50 --windows 11
tools/ab_roll.sh atm_b 1 --workload atmosmodd --steps 500 --warmup 50 --windows 11
CASK_AB_MODES="0 2" tools/ab_roll.sh cg_b 2 --workload G3_circuit --solver cg --steps 200 --warmup 20
CASK_AB_MODES="0 2" tools/ab_roll.sh bicg_b 2 --workload atmosmodd --solver bicg --steps 200 --warmup 20
# SCAN: the x window sharing the product area's LDS (8 instead of 4 workgroups per CU with a 2 048-entry window)
for wl in webbase2 webbase-1M; do
  tools/ab_env.sh scan_$wl 1 " -- --tile -1|CASK_HIP_SCAN_ALIAS=0 -- --tile 2048|CASK_HIP_SCAN_ALIAS=1 -- --tile 2048|CASK_HIP_SCAN_ALIAS=1 -- --tile 1024" --workload $wl --variant scan --wg 256 --items 8 --steps 500 --warmup 50 --windows 11
done
echo "== trsv"
timeout -k 10 600 python3 tools/bench_solvers.py trsv G3_circuit cant atmosmodd > gpurun_out/r05_trsv.jsonl 2> gpurun_out/r05_trsv.err; echo "trsv rc=$?"; cat gpurun_out/r05_trsv.jsonl | cut -c1-1500; tail -3 gpurun_out/r05_trsv.err
echo "== GPU suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r05_gputests_2.log 2>&1; echo "pytest rc=$?"; tail -50 gpurun_out/r05_gputests_2.log
