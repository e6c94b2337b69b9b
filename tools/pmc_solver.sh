#!/bin/bash
# L2 <-> fabric traffic of ONE SOLVER PASS: three counter passes (FETCH_SIZE, WRITE_SIZE, read
# requests by size; each its own rocprofv3 run with --kernel-trace only, program directly after `--`) of
# `bench.py --workload <w> --solver <s>`, plus the calibration kernel.  Output: gpurun_out/traffic_<w>_<s>_<tag>.json
#   tools/pmc_solver.sh <tag> <workload> <cg|bicg>
set -u
tag=$1; w=$2; sv=$3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
make build/membench > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE RDREQ; do
  ctr=$c
  if [ $c = RDREQ ]; then ctr="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; fi
  [ -n "$(find $out/pmc_${tag}_calib_$c -name '*counter_collection.csv' 2>/dev/null | head -n 1)" ] || rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${tag}_calib_$c -- $root/build/membench > /dev/null 2> $out/pmc_${tag}_calib_$c.err
  rm -rf $out/pmc_${tag}_${w}_${sv}_$c
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${tag}_${w}_${sv}_$c -- python3 $root/bench.py --workload $w --solver $sv --steps 40 --warmup 4 --windows 3 --no-cpu-baseline > /dev/null 2> $out/pmc_${tag}_${w}_${sv}_$c.err
done
python3 $root/tools/traffic_solver_summary.py $tag $w $sv
find $out -path "*pmc_${tag}_${w}_${sv}_*" -name "*.csv" -delete
