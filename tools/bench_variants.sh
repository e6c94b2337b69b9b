#!/bin/bash
# Every bench.py workload / launch form once (1 rank, RCCL at world 1, 4-rank dry runs sharing the GPU); run on the GPU box.
set -x
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/b1.json 2> gpurun_out/b1.err; tail -c 600 gpurun_out/b1.err
python bench.py --steps 20 --warmup 5 --workload webbase-1M --no-tune --cpu-seconds 2 > gpurun_out/b2.json 2> gpurun_out/b2.err; tail -c 600 gpurun_out/b2.err
python bench.py --steps 20 --warmup 5 --workload atmosmodd --solver bicg --cpu-seconds 2 > gpurun_out/b3.json 2> gpurun_out/b3.err; tail -c 600 gpurun_out/b3.err
python bench.py --steps 100 --warmup 5 --workload G3_circuit --solver cg --cpu-seconds 2 > gpurun_out/b4.json 2> gpurun_out/b4.err; tail -c 600 gpurun_out/b4.err
CASK_BENCH_FORCE_DIST=1 CASK_BENCH_EXCHANGE=all_gather MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 python bench.py --steps 20 --warmup 5 --workload webbase-1M --no-tune --no-cpu-baseline > gpurun_out/b5.json 2> gpurun_out/b5.err; tail -c 600 gpurun_out/b5.err
CASK_BENCH_FORCE_DIST=1 CASK_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 python bench.py --steps 20 --warmup 5 --workload atmosmodd --solver bicg --no-cpu-baseline > gpurun_out/b6.json 2> gpurun_out/b6.err; tail -c 600 gpurun_out/b6.err
CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 20 --warmup 5 --workload webbase-1M --no-tune --no-cpu-baseline > gpurun_out/b7.json 2> gpurun_out/b7.err; tail -c 600 gpurun_out/b7.err
CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 4 --steps 20 --warmup 5 --workload atmosmodd --solver bicg --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err; tail -c 600 gpurun_out/b8.err
python bench.py --steps 20 --warmup 5 --workload cant3 --cpu-seconds 2 > gpurun_out/b9.json 2> gpurun_out/b9.err; tail -c 600 gpurun_out/b9.err
for i in 1 2 3 4 5 6 7 8 9; do echo "== b$i"; grep "^{" gpurun_out/b$i.json | cut -c1-1500; done
