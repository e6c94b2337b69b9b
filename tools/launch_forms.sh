#!/bin/bash
# One box, one design point, the launch forms side by side (what does a step cost as a stream launch, as a graph node,
# with and without window events):  tools/launch_forms.sh <tag>
tag=${1:-x}
out=gpurun_out/launch_forms_${tag}.txt
: > $out
common="--gpus 1 --warmup 5 --no-others --no-cpu-baseline --variant merge --lanes 16 --tile 1024 --items 8 --wg 256"
run() {  # label, env, launch, steps
  env $2 python3 bench.py $common --launch $3 --steps $4 2>>gpurun_out/launch_forms_${tag}.err | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('%-22s K=%-5d R=%-4d median %.3f  mean %.3f  p10 %.3f  p90 %.3f  min %.3f max %.3f first %.3f us/step   wall %.3f' % ('$1', r['steps'], r['windows'], r['ms_per_step']*1e3, r['ms_per_step_mean']*1e3, r['ms_per_step_p10']*1e3, r['ms_per_step_p90']*1e3, r['ms_per_step_min']*1e3, r['ms_per_step_max']*1e3, r['ms_per_step_first_window']*1e3, r['host_wall_ms_per_step']*1e3))
" | tee -a $out
}
for rep in 1 2; do
  run "sequence(native ev)" "X=1" sequence 20
  run "sequence(torch ev)" "CASK_BENCH_TORCH_EVENTS=1" sequence 20
  run "graph" "X=1" graph 20
  run "graph" "X=1" graph 1000
  run "sequence(native ev)" "X=1" sequence 1000
done
