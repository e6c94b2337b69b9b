#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_host_entry_gpu.py -x -q -m gpu -s > gpurun_out/tc_pytest.log 2>&1 || { tail -40 gpurun_out/tc_pytest.log; exit 1; }
tail -5 gpurun_out/tc_pytest.log
timeout -k 10 900 python3 tools/host_path_rate.py cant G3_circuit > gpurun_out/host_entry.txt 2> gpurun_out/host_entry.err; cat gpurun_out/host_entry.txt; tail -3 gpurun_out/host_entry.err
timeout -k 10 600 python3 -m pytest tests/test_precond_gpu.py tests/test_nonfinite_gpu.py -x -q -m gpu -k "trsolve or schedules or ilu" > gpurun_out/tc_pytest2.log 2>&1; tail -5 gpurun_out/tc_pytest2.log
