#!/bin/bash
# kernel stats of tools/family_times.py for one matrix / design points (run on the GPU box)
#   tools/prof_family.sh <tag> <matrix> <spec> [spec ...]
set -u
tag=$1; mat=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "$@"; do
  i=$((i+1))
  d=$out/fam_${tag}_${mat}_$i
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/family_times.py $mat $spec > $d.json 2> $d.err
  echo "== $spec"; grep usec_cold $d.json | cut -c1-120
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:4]:
    print(f'  {r["Name"][:100]:100s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:8.2f} min_us={float(r["MinNs"])/1e3:8.2f}')
PY
  find $d -name "*kernel_trace.csv" -delete
done
