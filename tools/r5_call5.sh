#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tools/ab_env.sh prio 2 " -- |CASK_HIP_MERGE_PRIO=1 -- |CASK_HIP_MERGE_PRIO=3 -- " --no-tune --steps 1000 --warmup 100 --windows 11
tools/ab_env.sh prio_g3 1 " -- |CASK_HIP_MERGE_PRIO=1 -- |CASK_HIP_MERGE_PRIO=3 -- " --no-tune --workload G3_circuit --steps 500 --warmup 50 --windows 11
echo "== GPU suite"
SECONDS=0
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r05_gputests_5.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"; tail -40 gpurun_out/r05_gputests_5.log | cut -c1-300
echo "== default line"
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; echo "default rc=$? wall ${SECONDS}s"
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_r05.json').read().strip().splitlines()[-1])
print('headline %.3f us frac %.4f value %.1f first %.3f' % (r['ms_per_step']*1e3, r['roofline']['frac'], r['value'], r.get('ms_per_step_first_window',0)*1e3), r['config']['design_point'])
print('cpu_baseline', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
for o in r['config']['other_workloads']:
    print(o.get('workload','?')[:40], o.get('usec'), o.get('frac'), (o.get('design_point') or {}).get('variant'), (o.get('design_point') or {}).get('tile_width'), 'cpu', (o.get('cpu_baseline') or {}).get('value'), o.get('rows_wrong'), (o.get('solve_check') or {}).get('iterations'), o.get('seconds_in_bench'), o.get('error'))
PY
