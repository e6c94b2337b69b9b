#!/bin/bash
# Where a solver pass's time goes (run on the GPU box): rocprofv3 kernel trace of `bench.py --solver`, per kernel of the pass its
# average duration and the average gap to the launch before it (end -> start), over the timed passes.
#   tools/pass_timeline.sh <tag> <workload> <cg|bicg>
tag=$1; w=$2; sv=$3
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out; d=$out/timeline_${tag}_${w}_$sv
cd /tmp && export TMPDIR=/tmp
rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $root/bench.py --workload $w --solver $sv --steps 100 --warmup 10 --no-cpu-baseline > $d.json 2> $d.err || exit 1
python3 - $d <<'PY'
import csv, glob, sys, collections, statistics, json
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the steady part: the last 60 % of the dispatches
rows = rows[len(rows) * 2 // 5:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = b["Kernel_Name"].split("(")[0].replace("void caskhip::", "")[:50]
    dur[name].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
    gap[name].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
tot = 0.0
for k in dur:
    print(f"{k:52s} n={len(dur[k]):6d} dur {statistics.mean(dur[k])/1e3:7.2f} us (median {statistics.median(dur[k])/1e3:7.2f})  gap before {statistics.median(gap[k])/1e3:6.2f} us")
line = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])
print("bench line:", round(line["ms_per_step"] * 1e3, 2), "us per pass")
PY
find $d -name "*.csv" -delete
