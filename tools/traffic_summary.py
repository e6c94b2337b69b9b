#!/usr/bin/env python3
"""Turn the PMC passes of tools/profile_round.sh into gpurun_out/traffic_<workload>_<tag>.json
(copied to profiles/traffic_<workload>.json, which bench.py reads into roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB.  The calibration kernel (tools/membench.hip
k_oneshot<8,true>) reads exactly 12*nnz bytes with 16-byte-per-lane nontemporal
loads; the factor known_bytes / (FETCH_SIZE*1024) corrects gfx950's FETCH_SIZE
for that access shape (the guide measures exactly 2.0) and is applied to the
SpMV kernel's FETCH_SIZE.  WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import statistics
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "cant"
point = sys.argv[3] if len(sys.argv) > 3 else ""            # design point label ("" = AUTO): keys the output file
suffix = f"_{point}" if point else ""
root = Path(__file__).resolve().parent.parent
out = root / "gpurun_out"


def counter(dirpat, kernel_sub, name):
    vals = []
    for f in glob.glob(str(out / dirpat / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 4:]                      # drop warm-up launches
    return statistics.median(vals) if vals else None, len(vals)


CAL_NNZ = 4010891
fetch_cal, n1 = counter(f"pmc_{tag}_calib_FETCH_SIZE", "k_oneshot<8, true>", "FETCH_SIZE")
known = 12.0 * CAL_NNZ
factor = known / (fetch_cal * 1024.0) if fetch_cal else None
cal_sizes = {n: counter(f"pmc_{tag}_calib_RDREQ", "k_oneshot<8, true>", n)[0]
             for n in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
res = {"tag": tag, "calibration": {"kernel": "k_oneshot<8,true> (tools/membench.hip)", "known_bytes": known,
                                   "FETCH_SIZE_KiB": fetch_cal, "factor": factor, "samples": n1,
                                   "read_bytes_by_request_size": (round(32 * cal_sizes["TCC_EA0_RDREQ_32B_sum"] +
                                                                        64 * cal_sizes["TCC_EA0_RDREQ_64B_sum"] +
                                                                        128 * cal_sizes["TCC_EA0_RDREQ_128B_sum"])
                                                                  if all(v is not None for v in cal_sizes.values()) else None)}}
for kern in ("k_spmv_merge<", "k_spmv_scan_pad<", "k_spmv_scan<", "k_spmv_slice<", "k_spmv_vector2<", "k_spmv_vector<", "k_spmv_merge_wave<"):
    f, nf = counter(f"pmc_{tag}_{workload}{suffix}_FETCH_SIZE", kern, "FETCH_SIZE")
    w, nw = counter(f"pmc_{tag}_{workload}{suffix}_WRITE_SIZE", kern, "WRITE_SIZE")
    if f is None:
        continue
    res["kernel"] = kern.rstrip("<")
    res["FETCH_SIZE_KiB"] = f
    res["WRITE_SIZE_KiB"] = w
    res["samples"] = [nf, nw]
    if factor:
        res["hbm_bytes_per_launch"] = round(f * 1024.0 * factor + (w or 0) * 1024.0)
    # cross-check: L2 -> memory read requests by size (exact, no correction)
    sizes = {}
    for name, nbytes in (("TCC_EA0_RDREQ_32B_sum", 32), ("TCC_EA0_RDREQ_64B_sum", 64), ("TCC_EA0_RDREQ_128B_sum", 128)):
        v, _ = counter(f"pmc_{tag}_{workload}{suffix}_RDREQ", kern, name)
        if v is not None:
            sizes[name] = v
    if sizes:
        rd = sum(v * {"TCC_EA0_RDREQ_32B_sum": 32, "TCC_EA0_RDREQ_64B_sum": 64, "TCC_EA0_RDREQ_128B_sum": 128}[k]
                 for k, v in sizes.items())
        res["read_requests_by_size"] = sizes
        res["read_bytes_by_request_size"] = round(rd)
        res["hbm_bytes_per_launch_by_request_size"] = round(rd + (w or 0) * 1024.0)
    break
print(json.dumps(res, indent=1))
res["workload"] = workload
res["design_point"] = point or "auto"
(root / "gpurun_out" / f"traffic_{workload}{suffix}_{tag}.json").write_text(json.dumps(res, indent=1))
