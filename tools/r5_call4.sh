#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "== failing test first"
timeout -k 10 300 python -m pytest tests/test_bench_gpu.py -x -q -k "config5 or two_rank_dry" > gpurun_out/r05_t_config5.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/r05_t_config5.log | cut -c1-600
echo "== default line"
start=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default.json 2> gpurun_out/r05_default.err; echo "default rc=$? wall $(echo "$(date +%s.%N) - $start" | bc) s"
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r05_default.json').read().strip().splitlines()[-1])
print('headline %.3f us frac %.4f value %.1f' % (r['ms_per_step']*1e3, r['roofline']['frac'], r['value']), r['config']['design_point'])
print('cpu_baseline', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
for o in r['config']['other_workloads']:
    print(o.get('workload','?')[:40], o.get('usec'), o.get('frac'), (o.get('design_point') or {}).get('variant'), (o.get('design_point') or {}).get('tile_width'), 'cpu', (o.get('cpu_baseline') or {}).get('value'), o.get('rows_wrong'), (o.get('solve_check') or {}).get('iterations'), o.get('seconds_in_bench'), o.get('error'))
PY
echo "== GPU suite"
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r05_gputests_4.log 2>&1; echo "pytest rc=$?"; tail -45 gpurun_out/r05_gputests_4.log | cut -c1-400
