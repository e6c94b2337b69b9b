#!/bin/bash
# Round profile for the headline bench (run on the GPU box via gpurun):
#   tools/profile_round.sh <tag> [workload ...]     e.g. r01 cant webbase-1M
# 1. bench.py as the driver runs it              -> gpurun_out/bench_<tag>.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/prof_<tag>/
# 3. HBM traffic per workload: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes, for the SpMV
#    kernel and for a calibration kernel that streams a known byte count with the
#    same 16-byte access shape (tools/membench.hip k_oneshot), because gfx950's
#    FETCH_SIZE under-reports wide coalesced reads (MI355X_MICROARCH.md, HBM); and a third pass
#    with the read requests by size, which counts the bytes exactly.
set -u
tag=${1:-r01}
shift || true
workloads=${*:-cant}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
rm -rf $out/prof_$tag $out/pmc_${tag}_*          # a merged gpurun_out/ may still hold an earlier run's files
cd $root
make build/membench > /dev/null 2>&1
python3 bench.py > $out/bench_$tag.json 2> $out/bench_$tag.err
cat $out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 $root/bench.py --no-cpu-baseline --no-others > $out/prof_$tag.json 2> $out/prof_$tag.err
# three separate counter passes: FETCH_SIZE, WRITE_SIZE, and the L2->memory read requests split by size
# (32/64/128 bytes: an exact byte count that needs no correction; the cross-check of the corrected FETCH_SIZE)
for c in FETCH_SIZE WRITE_SIZE RDREQ; do
  ctr=$c
  if [ $c = RDREQ ]; then ctr="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; fi
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${tag}_calib_$c -- $root/build/membench > /dev/null 2> $out/pmc_${tag}_calib_$c.err
  for w in $workloads; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${tag}_${w}_$c -- python3 $root/bench.py --workload $w --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-tune --no-others > /dev/null 2> $out/pmc_${tag}_${w}_$c.err
  done
done
for w in $workloads; do python3 $root/tools/traffic_summary.py $tag $w; done
# gpurun copies back at most 64 MiB: keep the summaries, drop the per-dispatch traces
find $out -name "*counter_collection.csv" -delete
find $out -path "*pmc_*" -name "*kernel_trace.csv" -delete
find $out -path "*prof_*" -name "*kernel_trace.csv" -delete
# the README line of the kernel-stats CSV, generated from the CSV itself (copy BOTH into profiles/ together)
stats=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$stats" ]; then
  alg=$(python3 -c "import json;print(json.load(open('$out/bench_$tag.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null)
  python3 $root/tools/kernel_stats_line.py $stats $alg | tee $out/prof_${tag}_readme_line.txt
fi
