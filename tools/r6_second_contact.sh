#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_spmv_gpu.py tests/test_random_gpu.py -x -q -m gpu -k "slice or scan or random or long or full_size" > gpurun_out/sc_pytest1.log 2>&1 || { tail -30 gpurun_out/sc_pytest1.log; exit 1; }
tail -3 gpurun_out/sc_pytest1.log
bash tools/slice_probe.sh webbase2 webbase-1M > gpurun_out/sc_probe.log 2>&1 || { tail -20 gpurun_out/sc_probe.log; exit 1; }
cut -c1-200 gpurun_out/slice_probe.txt
# does the non-finite test see the round-5 mask?  (expected: failures under mul, none under the select)
CASK_HIP_TRSV_LANES_MASK=mul timeout -k 10 600 python3 -m pytest tests/test_nonfinite_gpu.py -q -m gpu -k trsolve > gpurun_out/sc_nonfinite_mul.log 2>&1; tail -12 gpurun_out/sc_nonfinite_mul.log | cut -c1-250
