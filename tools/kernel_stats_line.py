#!/usr/bin/env python3
"""The README line of a round's kernel-stats CSV, generated FROM the CSV so that the prose cannot drift from the file
(VERDICT r4 item 9):  tools/kernel_stats_line.py profiles/r05_bench_kernel_stats.csv [algorithmic bytes per launch]
Prints the headline kernel's average / min / calls and, with the byte count, the roofline fraction at 8 TB/s."""
import csv
import sys


def main():
    path = sys.argv[1]
    alg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = list(csv.DictReader(open(path)))
    spmv = [r for r in rows if "k_spmv_" in r["Name"]]
    spmv.sort(key=lambda r: -float(r["TotalDurationNs"]) if "TotalDurationNs" in r else -float(r["Percentage"]))
    r = spmv[0]
    name = r["Name"].split("(")[0].replace("void caskhip::", "")
    avg, mn, calls = float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, int(r["Calls"])
    line = f"`{name}` {avg:.3f} µs average over " + f"{calls:,}".replace(",", " ") + f" dispatches (min {mn:.2f})"
    if alg:
        line += f"; {alg} algorithmic bytes per launch / that = {alg / avg / 1e6:.2f} TB/s = frac {alg / avg / 1e6 / 8.0:.3f} of 8 TB/s"
    print(line)


if __name__ == "__main__":
    main()
