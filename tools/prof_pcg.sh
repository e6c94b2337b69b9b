#!/bin/bash
# per-kernel times of the PCG runs (tools/bench_solvers.py pcg noilu) under rocprofv3 --kernel-trace --stats
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_pcg
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/bench_solvers.py pcg noilu > $out/run.json 2> $out/run.err
cat $out/run.json
python3 - <<'PY'
import glob, csv, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_pcg"
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("  %-90s calls %6s avg %10.1f ns  pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
