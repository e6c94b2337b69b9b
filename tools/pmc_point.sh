#!/bin/bash
# HBM traffic of ONE design point of one workload (run on the GPU box): the three counter passes of
# tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE, read requests by size; each in its own rocprofv3 run, plus the
# calibration kernel) with bench.py forced to the point.  Output: gpurun_out/traffic_<workload>_<label>_<tag>.json
#   tools/pmc_point.sh <tag> <workload> <label> [bench.py design-point flags: --variant scan --tile 4096 ...]
set -u
tag=$1; w=$2; label=$3; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
make build/membench > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE RDREQ; do
  ctr=$c
  if [ $c = RDREQ ]; then ctr="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; fi
  # (the calibration of this tag is reused only while its counter file is there: tools/profile_round.sh deletes the CSVs it has summarised)
  [ -n "$(find $out/pmc_${tag}_calib_$c -name '*counter_collection.csv' 2>/dev/null | head -n 1)" ] || rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${tag}_calib_$c -- $root/build/membench > /dev/null 2> $out/pmc_${tag}_calib_$c.err
  rm -rf $out/pmc_${tag}_${w}_${label}_$c
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${tag}_${w}_${label}_$c -- python3 $root/bench.py --workload $w --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-others "$@" > /dev/null 2> $out/pmc_${tag}_${w}_${label}_$c.err
done
python3 $root/tools/traffic_summary.py $tag $w $label
find $out -path "*pmc_${tag}_${w}_${label}_*" -name "*.csv" -delete
