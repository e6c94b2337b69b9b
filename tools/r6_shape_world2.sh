#!/bin/bash
# r6: the two-rank shared-device rehearsal with the sharded solves in the SMALL shape (the default) and forced into the BIG one
cd "$GRAFT_REPO_ROOT" || exit 1
for shape in default big; do
  if [ $shape = big ]; then export CASK_HIP_SOLVER_SHAPE=big; else unset CASK_HIP_SOLVER_SHAPE; fi
  bash tools/world_dryrun.sh 2 > /dev/null 2>&1
  echo "== solver shape: $shape"; grep -E "exit status|appended" gpurun_out/world_dryrun.txt | cut -c1-200
done
