#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/exp4
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
for nt in 0 -1; do
  export CASK_SOLVER_NT=$nt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/solv_nt$nt -- python3 $root/tools/bench_solvers.py G3_circuit atmosmodd > $out/solv_nt$nt.json 2> $out/solv_nt$nt.err
  cat $out/solv_nt$nt.json
done
python3 - <<'PY'
import glob, csv, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/exp4"
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    print("==", f[len(root):])
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-70s calls %6s avg %10.1f ns  pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
