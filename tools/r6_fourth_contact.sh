#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_solvers_gpu.py tests/test_fused_gpu.py -x -q -m gpu > gpurun_out/fo_pytest.log 2>&1 || { tail -40 gpurun_out/fo_pytest.log; exit 1; }
tail -3 gpurun_out/fo_pytest.log
bash tools/bicg_dual_ab.sh > gpurun_out/fo_ab.log 2>&1; cat gpurun_out/bicg_dual_ab.txt; tail -3 gpurun_out/bicg_dual_ab.err
