#!/bin/bash
# Interleaved A/B of a solver line between the product library and another build (run on the GPU box):
#   tools/ab_solver.sh <tag> <other .so> <workload> <cg|bicg>     -> gpurun_out/ab_<tag>.txt
tag=$1; lib=$2; w=$3; sv=$4
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_$tag.txt
: > $out
for i in 1 2 3; do
  for which in new other; do
    if [ $which = other ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/$lib; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
    python3 $root/bench.py --no-cpu-baseline --workload $w --solver $sv --steps 200 --warmup 20 2>>$root/gpurun_out/ab_$tag.err | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('$which $i  %.3f us per pass  p10 %.3f p90 %.3f  iterations %s (oracle %s)' % (r['ms_per_step']*1e3, r['ms_per_step_p10']*1e3, r['ms_per_step_p90']*1e3, r['config']['solve_check']['iterations'], r['config']['solve_check']['oracle_iterations']))" | tee -a $out
  done
done
