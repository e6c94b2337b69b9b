#!/bin/bash
# rocprofv3 kernel statistics of one ILU(0) application per schedule (tools/trsv_bench.py); run on the GPU box.
#   bash tools/trsv_profile.sh <round tag> [matrix]
TAG=${1:-r03}; M=${2:-G3_circuit}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trsvprof -o trsv -- python3 $GRAFT_REPO_ROOT/tools/trsv_bench.py $M > $GRAFT_REPO_ROOT/gpurun_out/trsvprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/trsvprof.err
echo rc=$?
f=$(find $GRAFT_REPO_ROOT/gpurun_out/trsvprof -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trsv_kernel_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/trsvprof -name "*.csv" -size +1M -delete
