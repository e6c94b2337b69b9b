#!/bin/bash
# dev experiment: A/B of the headline bench between the previous commit's engine (build/base) and the current one
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/exp3
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/base/libcask_hip.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-tune > $out/bench_${which}_$i.json 2> $out/bench_${which}_$i.err
    python3 -c "import json,sys; d=json.load(open('$out/bench_${which}_$i.json')); print('$which $i', d['value'], d['roofline']['launch_usec'])"
  done
done
unset CASK_HIP_DIAGNOSTIC_LIB
timeout -k 10 300 python tools/bench_solvers.py > $out/solvers.json 2> $out/solvers.err; cat $out/solvers.json
