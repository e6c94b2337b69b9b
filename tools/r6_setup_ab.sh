#!/bin/bash
# r6: the fused set-up of a solve (k_copy2 + k_solver_residual) against the previous build (build/diag/libcask_hip_head.so): parity,
# then short solves (20-pass windows: the set-up is a tenth of them) and the 200-pass windows of the bench line.
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; out=$root/gpurun_out
timeout -k 10 600 python -m pytest tests/test_solvers_gpu.py tests/test_dist_gpu.py tests/test_p2p_gpu.py tests/test_host_cpp.py -x -q -m gpu > $out/setup_tests.log 2>&1 || { tail -30 $out/setup_tests.log; exit 1; }
tail -1 $out/setup_tests.log
for w in "atmosmodd bicg" "G3_circuit cg" "cant cg"; do set -- $w
for i in 1 2 3; do for which in new other; do
  if [ $which = other ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/diag/libcask_hip_head.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
  python3 bench.py --no-cpu-baseline --workload $1 --solver $2 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$1 $2 $which $i (20-pass windows)  %.3f us per pass' % (r['ms_per_step']*1e3))"
done; done; done
bash tools/ab_solver.sh setup_cg build/diag/libcask_hip_head.so G3_circuit cg || exit 1
bash tools/ab_solver.sh setup_bicg build/diag/libcask_hip_head.so atmosmodd bicg || exit 1
