#!/bin/bash
# Development tool (run on the GPU box via gpurun): quick interleaved A/B of the headline bench between
# build/base/libcask_hip.so and the current engine, after a subset of the GPU tests.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/abq
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests/test_spmv_gpu.py tests/test_fused_gpu.py tests/test_p2p_gpu.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/base/libcask_hip.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-tune > $out/bench_${which}_$i.json 2> $out/bench_${which}_$i.err
    python3 -c "import json,sys; d=json.load(open('$out/bench_${which}_$i.json')); print('$which $i', d['value'], d['roofline']['launch_usec'])"
  done
done
unset CASK_HIP_DIAGNOSTIC_LIB
for w in G3_circuit atmosmodd; do for which in base new; do
  if [ $which = base ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/base/libcask_hip.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
  timeout -k 10 300 python bench.py --workload $w --no-tune --no-cpu-baseline --steps 200 --warmup 20 > $out/bench_${w}_${which}.json 2>/dev/null; python3 -c "import json; d=json.load(open('$out/bench_${w}_${which}.json')); print('$w $which', d['value'], d['roofline']['launch_usec'])"; done; done
