#!/bin/bash
# round 5, third GPU call: kernarg preload / device kernarg A/B on the headline, the default line (timed), the GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R=$PWD
tools/ab_env.sh kernarg 3 " -- |CASK_HIP_DIAGNOSTIC_LIB=$R/build/libcask_hip_preload.so -- |HIP_FORCE_DEV_KERNARG=1 -- |HIP_FORCE_DEV_KERNARG=1 CASK_HIP_DIAGNOSTIC_LIB=$R/build/libcask_hip_preload.so -- " --no-tune --steps 1000 --warmup 100 --windows 11
tools/ab_env.sh kernarg20 2 " -- |CASK_HIP_DIAGNOSTIC_LIB=$R/build/libcask_hip_preload.so -- |HIP_FORCE_DEV_KERNARG=1 -- |HIP_FORCE_DEV_KERNARG=1 CASK_HIP_DIAGNOSTIC_LIB=$R/build/libcask_hip_preload.so -- " --no-tune --steps 20 --warmup 5
tools/ab_env.sh kernarg_g3 1 " -- |CASK_HIP_DIAGNOSTIC_LIB=$R/build/libcask_hip_preload.so -- |HIP_FORCE_DEV_KERNARG=1 -- " --no-tune --workload G3_circuit --steps 500 --warmup 50 --windows 11
echo "== default line"
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default.json 2> gpurun_out/r05_default.err; echo "default rc=$?"; grep -E "Elapsed|Maximum resident" gpurun_out/r05_default.err
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r05_default.json').read().strip().splitlines()[-1])
print('headline %.3f us frac %.4f value %.1f' % (r['ms_per_step']*1e3, r['roofline']['frac'], r['value']), r['config']['design_point'])
print('cpu_baseline', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
for o in r['config']['other_workloads']:
    print(o.get('workload','?')[:40], o.get('usec'), o.get('frac'), (o.get('design_point') or {}).get('variant'), (o.get('design_point') or {}).get('tile_width'), 'cpu', (o.get('cpu_baseline') or {}).get('value'), o.get('rows_wrong'), (o.get('solve_check') or {}).get('iterations'), o.get('seconds_in_bench'), o.get('error'))
PY
echo "== GPU suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r05_gputests_3.log 2>&1; echo "pytest rc=$?"; tail -35 gpurun_out/r05_gputests_3.log
