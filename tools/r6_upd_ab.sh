#!/bin/bash
# r6: the update launches of a classic solver pass with all of a lane's pairs requested ahead of the scalars (UPD_AHEAD = 4),
# against the previous build (build/diag/libcask_hip_lines.so: only the first pair ahead) -- parity of everything that solves,
# then interleaved A/Bs of the CG and BiCG passes on one box, then the per-kernel durations (rocprofv3).
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; out=$root/gpurun_out
timeout -k 10 600 python -m pytest tests/test_solvers_gpu.py tests/test_dist_gpu.py tests/test_p2p_gpu.py tests/test_push_gpu.py tests/test_precond_gpu.py tests/test_fused_gpu.py tests/test_nonfinite_gpu.py -x -q -m gpu > $out/upd_tests.log 2>&1 || { tail -30 $out/upd_tests.log; exit 1; }
tail -1 $out/upd_tests.log
bash tools/ab_solver.sh upd_cg build/diag/libcask_hip_lines.so G3_circuit cg || exit 1
bash tools/ab_solver.sh upd_bicg build/diag/libcask_hip_lines.so atmosmodd bicg || exit 1
bash tools/pass_timeline.sh r06upd G3_circuit cg && bash tools/pass_timeline.sh r06upd atmosmodd bicg
