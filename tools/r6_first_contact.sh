#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_spmv_gpu.py tests/test_random_gpu.py -x -q -m gpu -k "slice or vector or removed or scan or random or long or full_size" > gpurun_out/fc_pytest1.log 2>&1 || { tail -30 gpurun_out/fc_pytest1.log; exit 1; }
tail -3 gpurun_out/fc_pytest1.log
bash tools/slice_probe.sh webbase2 webbase-1M > gpurun_out/fc_probe.log 2>&1 || { tail -20 gpurun_out/fc_probe.log; exit 1; }
cat gpurun_out/slice_probe.txt
bash tools/lanes_mask_ab.sh > gpurun_out/fc_lanes.log 2>&1 || { tail -20 gpurun_out/fc_lanes.log; exit 1; }
cat gpurun_out/lanes_mask_ab.txt
timeout -k 10 600 python3 -m pytest tests/test_nonfinite_gpu.py -q -m gpu > gpurun_out/fc_pytest2.log 2>&1; tail -40 gpurun_out/fc_pytest2.log
