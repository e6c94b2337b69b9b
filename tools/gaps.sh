#!/bin/bash
# Development probe (run via gpurun): distribution of the idle gap between consecutive SpMV launches inside the
# bench's timed HIP graph, from rocprofv3's kernel trace (start of launch i+1 minus end of launch i).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/gaps
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --steps 400 --warmup 40 --no-tune --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - "$out" <<'PY'
import csv, glob, sys, statistics
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_spmv_merge" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps, durs = [], []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if 0 <= g < 20000:                      # consecutive launches of one graph replay
        gaps.append(g); durs.append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
def pct(v, q): return sorted(v)[min(len(v) - 1, int(q * len(v)))]
print("launches", len(rows), "gaps considered", len(gaps))
print("all pairs: gap ns p10 %.0f median %.0f p90 %.0f ; duration ns p10 %.0f median %.0f p90 %.0f" % (
    pct(gaps, .1), pct(gaps, .5), pct(gaps, .9), pct(durs, .1), pct(durs, .5), pct(durs, .9)))
# per chunk of 400 consecutive launches (one graph replay each, once the tune is over)
for k in range(max(0, len(gaps) - 3600), len(gaps) - 399, 400):
    g, d = gaps[k:k + 400], durs[k:k + 400]
    print("  launches %5d..: gap median %5.0f  duration median %5.0f  sum/launch %5.0f" % (
        k, pct(g, .5), pct(d, .5), (sum(g) + sum(d)) / 400.0))
PY
find $out -name "*kernel_trace.csv" -delete
