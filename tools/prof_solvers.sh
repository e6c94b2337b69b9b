#!/bin/bash
# Per-kernel durations of the solver passes (run on the GPU box): rocprofv3 kernel stats of
# tools/bench_solvers.py for one matrix, in composed (1) and classic (2) mode.
#   tools/prof_solvers.sh <tag> <matrix> [modes...]
set -u
tag=${1:-r02}; mat=${2:-G3_circuit}; shift; shift
modes=${*:-1 2}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in $modes; do
  d=$out/solv_${tag}_${mat}_m$m
  rm -rf $d
  CASK_HIP_SOLVER_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/bench_solvers.py $mat > $d.json 2> $d.err
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== mode $m"; cat $d.json | cut -c1-260
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:10]:
    print(f'{r["Name"][:110]:110s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:8.2f} min_us={float(r["MinNs"])/1e3:8.2f}')
PY
  find $d -name "*kernel_trace.csv" -delete
done
