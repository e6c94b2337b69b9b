#!/bin/bash
# A/B of the lane-group walk's DPP mask: select on the moved value (r6, the engine's) against the
# round-5 multiplication by 0.0 (CASK_HIP_TRSV_LANES_MASK=mul), interleaved, one ILU(0) application on the cant-like
# factors (the lane-group walk's matrix) and on G3_circuit-like (walk2: a control that must not move).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/lanes_mask_ab.txt
: > $OUT
for rep in 1 2 3; do
  for arm in select mul; do
    for m in cant; do
      if [ $arm = mul ]; then export CASK_HIP_TRSV_LANES_MASK=mul; else unset CASK_HIP_TRSV_LANES_MASK; fi
      echo -n "[$arm] rep $rep " >> $OUT
      timeout -k 10 200 python3 tools/trsv_bench.py $m ilu0 2>/dev/null | cut -c1-200 >> $OUT || exit 1
    done
  done
done
cat $OUT
