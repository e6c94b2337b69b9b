#!/bin/bash
# Interleaved A/B of one bench.py command between the product library and a diagnostic build (run on the GPU box):
#   tools/ab_lib.sh <tag> <diag .so> [bench.py flags...]     -> gpurun_out/ab_<tag>.txt
tag=$1; lib=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_$tag.txt
: > $out
for i in 1 2 3; do
  for which in base diag; do
    if [ $which = diag ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/$lib; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
    python3 $root/bench.py --no-cpu-baseline --no-others "$@" 2>>$root/gpurun_out/ab_$tag.err | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print('$which $i  %.3f us  p10 %.3f p90 %.3f  wrong %s  %s' % (r['ms_per_step']*1e3, r['ms_per_step_p10']*1e3, r['ms_per_step_p90']*1e3, r['config']['rows_wrong_vs_oracle_all_ranks'], r['config']['design_point']['variant']))" | tee -a $out
  done
done
