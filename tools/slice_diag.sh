#!/bin/bash
# development: the two kinds of workgroup of a SLICE launch timed apart (CASK_HIP_SLICE_DIAG = 1: the long rows' SCAN blocks
# alone, 2: the slices alone; results are incomplete by construction), next to the whole launch and to SCAN
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/slice_diag.txt
: > $OUT
for m in ${@:-webbase2}; do
  for K in 1 2 4; do
    for diag in 0 1 2; do
      echo -n "K=$K diag=$diag " >> $OUT
      CASK_HIP_SLICE_DIAG=$diag timeout -k 10 300 python3 tools/family_times.py $m "variant=slice,lanes_per_row=$K,wg_size=256,items_per_thread=8,tile_width=2048" 2>/dev/null | cut -c1-140 >> $OUT
    done
  done
  timeout -k 10 300 python3 tools/family_times.py $m 'variant=scan,wg_size=256,items_per_thread=8,tile_width=2048' 2>/dev/null | cut -c1-140 >> $OUT
done
cat $OUT
