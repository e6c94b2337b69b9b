#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "== build() + smoke()"
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
echo "== GPU suite (driver form)"
SECONDS=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_final.log 2>&1; echo "pytest rc=$? wall ${SECONDS}s"; tail -3 gpurun_out/r05_gputests_final.log
echo "== driver command"
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r05_final.json 2> gpurun_out/bench_r05_final.err; echo "rc=$? wall ${SECONDS}s"
python3 -c "
import json
r=json.loads(open('gpurun_out/bench_r05_final.json').read().strip().splitlines()[-1])
print('headline %.3f us frac %.4f value %.1f first %.3f' % (r['ms_per_step']*1e3, r['roofline']['frac'], r['value'], r.get('ms_per_step_first_window',0)*1e3))
for o in r['config']['other_workloads']: print(o['workload'][:34], o['usec'], o['frac'], o.get('traffic'), (o.get('cpu_baseline') or {}).get('value'), o.get('rows_wrong'), (o.get('solve_check') or {}).get('iterations'))
"
