#!/bin/bash
# rocprofv3 kernel statistics of ILU(0) applications on the cant-like factors (default schedules):
#   tools/trsv_prof.sh <tag>   ->  gpurun_out/<tag>_trsv_kernel_stats.csv
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
tag=${1:-x}
mkdir -p $out && rm -rf $out/prof_trsv_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_trsv_$tag -- python3 $root/tools/trsv_bench.py cant ilu0_unit > $out/prof_trsv_$tag.json 2> $out/prof_trsv_$tag.err
echo "rc=$?"
f=$(find $out/prof_trsv_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_trsv_kernel_stats.csv && head -8 $out/${tag}_trsv_kernel_stats.csv | cut -c1-200
find $out/prof_trsv_$tag -name "*kernel_trace.csv" -delete
tail -1 $out/prof_trsv_$tag.json | cut -c1-200
