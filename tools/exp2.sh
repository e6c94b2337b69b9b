#!/bin/bash
# dev experiment: GPU tests, bench, 2-rank dry runs on one device (in-kernel halo vs pull), solvers
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/exp2
mkdir -p $out
cd $root
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json
for mode in fused pull; do
  if [ $mode = pull ]; then export CASK_BENCH_NO_FUSED_HALO=1; else unset CASK_BENCH_NO_FUSED_HALO; fi
  CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
     --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-tune > $out/bench2_$mode.json 2> $out/bench2_$mode.err
  echo "2-rank $mode rc=$?"; cat $out/bench2_$mode.json; tail -3 $out/bench2_$mode.err
done
unset CASK_BENCH_NO_FUSED_HALO
timeout -k 10 300 python tools/bench_solvers.py > $out/solvers.json 2> $out/solvers.err; cat $out/solvers.json
