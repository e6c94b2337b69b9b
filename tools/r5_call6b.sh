#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1
t() { SECONDS=0; "$@" > gpurun_out/r05_t.out 2> gpurun_out/r05_t.err; echo "rc=$? ${SECONDS}s :: $*"; grep "^\[bench\]" gpurun_out/r05_t.err | tail -12 | cut -c1-200; }
echo "== (a) halo fault, cant weak, 2 ranks"
CASK_FAULT_STALE_HALO=halo CASK_SELFCHECK_EXCHANGES=12 t python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 2 --no-tune --copies 2 --no-cpu-baseline --no-others --preroll-ms 40
echo "== (b) push fault, webbase/8, 3 ranks"
CASK_FAULT_STALE_HALO=push CASK_BENCH_SHRINK=8 CASK_SELFCHECK_EXCHANGES=12 t python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 3 --steps 10 --warmup 2 --workload webbase-1M --no-tune --copies 2 --no-cpu-baseline --preroll-ms 40
echo "== (c) all faults, atm/8 bicg, 2 ranks, peer allreduce"
CASK_FAULT_STALE_HALO=1 CASK_PEER_ALLREDUCE=1 CASK_BENCH_SHRINK=8 CASK_SELFCHECK_EXCHANGES=12 t python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 10 --warmup 2 --workload atmosmodd --solver bicg --no-cpu-baseline --preroll-ms 40
echo "== (c') the same at full size, 50 exchanges"
CASK_FAULT_STALE_HALO=1 CASK_PEER_ALLREDUCE=1 t python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 2 --steps 10 --warmup 2 --workload atmosmodd --solver bicg --no-cpu-baseline --preroll-ms 40
unset CASK_BENCH_SHARE_DEVICE CASK_BENCH_BACKEND MASTER_ADDR
python3 tools/dse_evidence.py r05 gpurun_out/r05_dse_out.json cant G3_circuit webbase-1M webbase2 atmosmodd > gpurun_out/r05_dse_evidence.log 2>&1; echo "dse_evidence rc=$?"; tail -8 gpurun_out/r05_dse_evidence.log | cut -c1-300
bash tools/pmc_solver.sh r05 G3_circuit cg > gpurun_out/r05_pmc_cg.log 2>&1; echo "pmc cg rc=$?"; tail -3 gpurun_out/r05_pmc_cg.log | cut -c1-300
bash tools/pmc_solver.sh r05 atmosmodd bicg > gpurun_out/r05_pmc_bicg.log 2>&1; echo "pmc bicg rc=$?"; tail -3 gpurun_out/r05_pmc_bicg.log | cut -c1-300
ls gpurun_out | grep -i "traffic.*r05\|r05_dse"
