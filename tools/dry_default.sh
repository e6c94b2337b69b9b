#!/bin/bash
# the driver's command form at N ranks on a 1-GPU box (ranks share the GPU, gloo control plane): `python bench.py --gpus N`
# starts its own ranks; prints the headline and the appended workloads
cd $GRAFT_REPO_ROOT
export CASK_BENCH_SHARE_DEVICE=1 CASK_BENCH_BACKEND=gloo
N=${1:-4}
( time python bench.py --gpus $N --steps 20 --warmup 5 --no-cpu-baseline ) 2>gpurun_out/dry_default.err | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
r=json.loads(ls[-1]); c=r['config']
print(r['n_gpus'], r['value'], r['ms_per_step'], c['exchange'][:50], c['rows_wrong_vs_oracle_all_ranks'])
for o in c.get('other_workloads', []): print('  ', o.get('config'), o.get('usec'), o.get('frac'), o.get('rows_wrong'), (o.get('solve_check') or {}).get('iterations'), str(o.get('exchange'))[:40], o.get('collectives'), o.get('error'), o.get('seconds_in_bench'))
"
grep -v "amdgpu.ids\|socket.cpp" gpurun_out/dry_default.err | tail -6
