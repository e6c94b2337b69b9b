#!/bin/bash
# dev experiment: kernel-level breakdown of the solvers (NT vs cached loads) and of the webbase-like SpMV
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/exp1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for nt in 0 -1; do
  export CASK_SOLVER_NT=$nt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/solv_nt$nt -- python3 $root/tools/bench_solvers.py G3_circuit atmosmodd > $out/solv_nt$nt.json 2> $out/solv_nt$nt.err
  cat $out/solv_nt$nt.json
done
unset CASK_SOLVER_NT
for w in webbase-1M G3_circuit atmosmodd; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_$w -- python3 $root/bench.py --workload $w --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_$w.json 2> $out/bench_$w.err
  cat $out/bench_$w.json
done
python3 - <<'PY'
import glob, csv, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/exp1"
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    print("==", f[len(root):])
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-70s calls %6s avg %10.1f ns  pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
