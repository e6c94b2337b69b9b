#!/bin/bash
# config-4 step at world 1 (every collective / exchange launch really issued): what the exchange adds to the product
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { MASTER_PORT=$((29800 + RANDOM % 100)) python bench.py --workload webbase-1M --steps 400 --warmup 40 --no-cpu-baseline --no-others "${@:2}" 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', r['ms_per_step']*1e3, 'us  ', r['config']['exchange'][:70], r['config']['design_point']['variant'], r['config']['rows_wrong_vs_oracle_all_ranks'])"; }
run "no exchange      " 
CASK_BENCH_FORCE_DIST=1 CASK_BENCH_EXCHANGE=push run "push             "
CASK_BENCH_FORCE_DIST=1 CASK_BENCH_EXCHANGE=all_gather run "rccl (native)    "
CASK_BENCH_FORCE_DIST=1 CASK_BENCH_EXCHANGE=all_gather CASK_NO_NATIVE_RCCL=1 run "rccl (torch)     "
