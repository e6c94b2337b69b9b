#!/bin/bash
# N-rank dry runs of every bench.py workload on a 1-GPU box (ranks share the GPU, gloo control plane), with the DSE on.
cd $GRAFT_REPO_ROOT
export CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 "${@:2}" 2>gpurun_out/dry.err | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('NO JSON'); sys.exit()
r=json.loads(ls[-1]); c=r['config']
print(r['n_gpus'], r['metric'][:40], r['value'], r['scaling'], c['exchange'][:60], c.get('rows_wrong_vs_oracle_all_ranks'), c.get('launch'), (c.get('tune') or {}).get('points'), (c.get('solve_check') or {}).get('iterations'))" || tail -5 gpurun_out/dry.err; }
run 4 --steps 20 --warmup 5 --copies 2 --no-cpu-baseline
run 3 --steps 20 --warmup 5 --copies 2 --no-cpu-baseline --workload G3_circuit
run 4 --steps 20 --warmup 5 --copies 2 --no-cpu-baseline --workload atmosmodd
run 4 --steps 20 --warmup 5 --copies 2 --no-cpu-baseline --workload webbase-1M
run 3 --steps 20 --warmup 5 --no-cpu-baseline --workload G3_circuit --solver cg
CASK_BENCH_EXCHANGE=all_gather run 2 --steps 20 --warmup 5 --no-cpu-baseline --workload atmosmodd --solver bicg
CASK_BENCH_EXCHANGE=p2p run 2 --steps 20 --warmup 5 --copies 2 --no-cpu-baseline --workload webbase-1M
tail -3 gpurun_out/dry.err
