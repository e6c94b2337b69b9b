#!/bin/bash
# Round-5 A/B of the lean merge kernel's forms on one box, interleaved (run on the GPU box via gpurun):
#   CASK_HIP_MERGE_ROLL = 0 plain | 1 rolling row sums | 2 products aliased over the x window | 3 both
#   tools/ab_roll.sh <tag> <reps> [bench.py flags...]     -> gpurun_out/ab_roll_<tag>.txt
tag=$1; reps=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab_roll_$tag.txt
: > $out
for i in $(seq 1 $reps); do
  for mode in ${CASK_AB_MODES:-0 1 2 3}; do
    CASK_HIP_MERGE_ROLL=$mode timeout -k 10 240 python3 $root/bench.py --no-cpu-baseline --no-others --no-tune "$@" 2>>$root/gpurun_out/ab_roll_$tag.err | tail -n 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
dp=r['config']['design_point']
print('roll $mode rep $i  %.3f us  p10 %.3f p90 %.3f  first %.3f  wrong %s  %s w%s i%s t%s lds %s grid %s' % (r['ms_per_step']*1e3, r['ms_per_step_p10']*1e3, r['ms_per_step_p90']*1e3, r.get('ms_per_step_first_window',0)*1e3, r['config'].get('rows_wrong_vs_oracle_all_ranks', r['config'].get('solve_check')), dp['variant'], dp['wg_size'], dp['items_per_thread'], dp['tile_width'], r['config'].get('lds_bytes'), r['config'].get('grid')))" | tee -a $out
  done
done
