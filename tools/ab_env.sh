#!/bin/bash
# Interleaved A/B of bench.py under several arms on one box (run on the GPU box via gpurun):
#   tools/ab_env.sh <tag> <reps> "<arm>|<arm>|..." [common bench.py flags...]   -> gpurun_out/ab_env_<tag>.txt
# an arm is "ENV1=a ENV2=b -- extra bench.py flags" (either side may be empty)
tag=$1; reps=$2; arms=$3; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_env_$tag.txt
: > $out
IFS='|' read -ra ARMS <<< "$arms"
for i in $(seq 1 $reps); do
  for arm in "${ARMS[@]}"; do
    envs="${arm%%--*}"; extra=""
    if [[ "$arm" == *--* ]]; then extra="--${arm#*--}"; extra="${extra#-- }"; fi
    env $envs timeout -k 10 240 python3 $root/bench.py --no-cpu-baseline --no-others "$@" $extra 2>>$root/gpurun_out/ab_env_$tag.err | tail -n 1 | ARM="$arm" python3 -c "
import json,sys,os
r=json.loads(sys.stdin.read())
dp=r['config']['design_point']
print('[%s] rep $i  %.3f us  p10 %.3f p90 %.3f  wrong %s  %s w%s i%s t%s far%s lds %s grid %s' % (os.environ['ARM'].strip(), r['ms_per_step']*1e3, r.get('ms_per_step_p10',0)*1e3, r.get('ms_per_step_p90',0)*1e3, r['config'].get('rows_wrong_vs_oracle_all_ranks', r['config'].get('solve_check')), dp['variant'], dp['wg_size'], dp['items_per_thread'], dp['tile_width'], dp['far_columns'], r['config'].get('lds_bytes'), r['config'].get('grid')))" | tee -a $out
  done
done
