#!/bin/bash
# Development tool (run on the GPU box via gpurun): GPU tests, then an interleaved A/B of the headline
# bench between a reference build of the engine (build/base/libcask_hip.so, e.g. the previous commit built
# in a git worktree) and the current one -- box-to-box variance is 2-3 %, an A/B on one box resolves 0.5 % --
# then the solver benchmarks under rocprofv3 --kernel-trace --stats.
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ab
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
for i in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export CASK_HIP_DIAGNOSTIC_LIB=$root/build/base/libcask_hip.so; else unset CASK_HIP_DIAGNOSTIC_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-tune > $out/bench_${which}_$i.json 2> $out/bench_${which}_$i.err
    python3 -c "import json,sys; d=json.load(open('$out/bench_${which}_$i.json')); print('$which $i', d['value'], d['roofline']['launch_usec'])"
  done
done
unset CASK_HIP_DIAGNOSTIC_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/solv -- python3 $root/tools/bench_solvers.py > $out/solv.json 2> $out/solv.err
cat $out/solv.json
python3 - <<'PY'
import glob, csv, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/ab"
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    print("==", f[len(root):])
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-70s calls %6s avg %10.1f ns  pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
