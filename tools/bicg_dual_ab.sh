#!/bin/bash
# A/B of the BiCG pass: the A and A^T products in ONE launch (r6, k_spmv_merge_dual) against the two-launch pass
# (CASK_HIP_NO_DUAL=1), interleaved, bench.py's solver line on the atmosmodd-like system (config 5 at one GPU).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/bicg_dual_ab.txt
: > $out
for i in 1 2 3; do
  for arm in dual two; do
    if [ $arm = two ]; then export CASK_HIP_NO_DUAL=1; else unset CASK_HIP_NO_DUAL; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload atmosmodd --solver bicg --steps 200 --warmup 20 2>>gpurun_out/bicg_dual_ab.err | tail -n 1 | ARM=$arm python3 -c "
import json,sys,os
r=json.loads(sys.stdin.read())
print('[%s] rep $i  %.3f us per pass  p10 %.3f p90 %.3f  iterations %s (oracle %s)  frac %.4f' % (os.environ['ARM'], r['ms_per_step']*1e3, r['ms_per_step_p10']*1e3, r['ms_per_step_p90']*1e3, r['config']['solve_check']['iterations'], r['config']['solve_check']['oracle_iterations'], r['roofline']['frac']))" | tee -a $out
  done
done
for i in 1 2; do
  for arm in new; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload G3_circuit --solver cg --steps 200 --warmup 20 2>>gpurun_out/bicg_dual_ab.err | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('[cg G3_circuit] rep $i  %.3f us per pass  iterations %s (oracle %s)  frac %.4f' % (r['ms_per_step']*1e3, r['config']['solve_check']['iterations'], r['config']['solve_check']['oracle_iterations'], r['roofline']['frac']))" | tee -a $out
  done
done
