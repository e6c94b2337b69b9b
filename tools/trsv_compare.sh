#!/bin/bash
# one ILU(0) application under the schedules of the triangular solves (tools/trsv_bench.py)
cd $GRAFT_REPO_ROOT
for m in G3_circuit atmosmodd cant; do
  for mode in default levels packed walk2 lanes; do CASK_HIP_TRSV=$mode python tools/trsv_bench.py $m 2>/dev/null | cut -c1-220; done
done
