#!/bin/bash
# r6: deferred solver checkpoints (DeferredFlags, cask_hip.hip) against the previous build (build/diag/libcask_hip_head.so): parity
# of everything that solves, interleaved A/Bs of the passes in 200-pass windows, then the driver's form (20-pass windows).
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; out=$root/gpurun_out
timeout -k 10 600 python -m pytest tests/test_solvers_gpu.py tests/test_dist_gpu.py tests/test_p2p_gpu.py tests/test_precond_gpu.py tests/test_host_cpp.py -x -q -m gpu > $out/defer_tests.log 2>&1 || { tail -30 $out/defer_tests.log; exit 1; }
tail -1 $out/defer_tests.log
bash tools/ab_solver.sh defer_cg build/diag/libcask_hip_head.so G3_circuit cg || exit 1
bash tools/ab_solver.sh defer_bicg build/diag/libcask_hip_head.so atmosmodd bicg || exit 1
bash tools/ab_solver.sh defer_cant build/diag/libcask_hip_head.so cant cg || exit 1
for i in 1 2; do for which in new other; do
  if [ $which = other ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/diag/libcask_hip_head.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
  python3 bench.py --no-cpu-baseline --workload atmosmodd --solver bicg --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('atmosmodd bicg $which $i (20-pass windows)  %.3f us per pass' % (r['ms_per_step']*1e3))"
done; done
