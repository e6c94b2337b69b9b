#!/bin/bash
# Sharded-solver collectives at world 1 on the nccl backend: the engine's native RCCL calls against torch.distributed callbacks.
cd $GRAFT_REPO_ROOT
export CASK_BENCH_FORCE_DIST=1 CASK_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for native in 1 0; do
  if [ $native = 0 ]; then export CASK_NO_NATIVE_RCCL=1; fi
  for w in "atmosmodd bicg" "G3_circuit cg"; do
    set -- $w
    MASTER_PORT=$((29600 + RANDOM % 200)) python bench.py --steps 200 --warmup 20 --workload $1 --solver $2 --no-cpu-baseline 2>gpurun_out/rccl_cmp.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']
print('native=$native', '$1', r['ms_per_step']*1e3, 'us/pass; host wall', r['host_wall_ms_per_step']*1e3, c['collectives'], c['exchange'][:40], c['solve_check']['iterations'], c['solve_check']['oracle_iterations'])"
  done
done
tail -3 gpurun_out/rccl_cmp.err
