#!/bin/bash
# sharded solver pass at world 1 with every reduction really issued: peer-store all-reduce vs RCCL vs none
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { label=$1; shift; MASTER_PORT=$((29800 + RANDOM % 100)) python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | LABEL="$label" python -c "
import json,sys,os
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']
print(os.environ['LABEL'], round(r['ms_per_step']*1e3,2), 'us  ', c['collectives'][:60], c['solve_check']['iterations'], c.get('pass_form'), c['exchange'][:30])"; }
for w in "atmosmodd --solver bicg" "G3_circuit --solver cg"; do
run "$w none      " --workload $w
CASK_BENCH_FORCE_DIST=1 CASK_FORCE_COLLECTIVES=1 CASK_PEER_ALLREDUCE=1 CASK_BENCH_EXCHANGE=p2p run "$w peer store" --workload $w
CASK_BENCH_FORCE_DIST=1 CASK_FORCE_COLLECTIVES=1 CASK_BENCH_EXCHANGE=p2p run "$w rccl      " --workload $w
done
