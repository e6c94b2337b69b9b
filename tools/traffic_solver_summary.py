#!/usr/bin/env python3
"""PMC passes of tools/pmc_solver.sh -> gpurun_out/traffic_<workload>_<solver>_<tag>.json (copied to
profiles/traffic_<workload>_<solver>.json, which bench.py reads into the solver lines' roofline.traffic).

Per kernel of a pass (the product launches, the update launches): the median counter value per dispatch and the
number of dispatches per pass (k_*_update_r runs exactly once per pass); bytes per pass = the sum.  FETCH_SIZE is
corrected with the factor the calibration kernel gives in the same session (gfx950 tallies a 128-byte request as 64
bytes: MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is taken as is; the read requests by size need no
correction and are the cross-check.  These counters sit on the L2's fabric side: Infinity-Cache hits are INCLUDED, so
for a working set under 256 MiB this is L2 <-> fabric traffic, an upper bound on what reached HBM.
"""
import csv
import glob
import json
import statistics
import sys
from pathlib import Path

tag, workload, solver = sys.argv[1], sys.argv[2], sys.argv[3]
root = Path(__file__).resolve().parent.parent
out = root / "gpurun_out"
PASS_KERNELS = ("k_spmv_merge<", "k_spmv_scan<", "k_cg_update_r", "k_cg_update_px", "k_bicg_update_r", "k_bicg_update_px",
                "k_dot_partial", "k_dot_final", "k_sum_to_scalars", "k_spmv_long", "k_spmv_fixup")


def per_kernel(dirpat, name):
    vals = {}
    for f in glob.glob(str(out / dirpat / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = next((p for p in PASS_KERNELS if p in r["Kernel_Name"]), None)
            if k:
                vals.setdefault(k.rstrip("<"), []).append(float(r["Counter_Value"]))
    return vals


def calib(name):
    v = []
    for f in glob.glob(str(out / f"pmc_{tag}_calib_{name if name != 'RDREQ' else 'RDREQ'}" / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_oneshot<8, true>" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                v.append(float(r["Counter_Value"]))
    v = v[len(v) // 4:]
    return statistics.median(v) if v else None


CAL_NNZ = 4010891
fetch_cal = calib("FETCH_SIZE")
factor = 12.0 * CAL_NNZ / (fetch_cal * 1024.0) if fetch_cal else None
fetch = per_kernel(f"pmc_{tag}_{workload}_{solver}_FETCH_SIZE", "FETCH_SIZE")
write = per_kernel(f"pmc_{tag}_{workload}_{solver}_WRITE_SIZE", "WRITE_SIZE")
rd = {n: per_kernel(f"pmc_{tag}_{workload}_{solver}_RDREQ", n)
      for n in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
once = next((k for k in fetch if k.endswith("update_r")), None)
res = {"tag": tag, "workload": workload, "solver": solver,
       "calibration": {"kernel": "k_oneshot<8,true> (tools/membench.hip)", "known_bytes": 12.0 * CAL_NNZ,
                       "FETCH_SIZE_KiB": fetch_cal, "factor": factor},
       "kernels": {}, "note": "L2 <-> fabric bytes per solver pass (Infinity-Cache hits included: an upper bound on HBM bytes)"}
total, total_by_size = 0.0, 0.0
if once and factor:
    passes = len(fetch[once])
    for k, v in sorted(fetch.items()):
        per_pass = len(v) / passes
        if per_pass < 0.5:                                   # set-up kernels (the first residual, the check solve's tail)
            continue
        tail = v[len(v) // 4:]
        f_kib = statistics.median(tail)
        w = write.get(k, [])
        w_kib = statistics.median(w[len(w) // 4:]) if w else 0.0
        by_size = sum(nb * statistics.median(rd[n][k][len(rd[n][k]) // 4:]) for n, nb in
                      (("TCC_EA0_RDREQ_32B_sum", 32), ("TCC_EA0_RDREQ_64B_sum", 64), ("TCC_EA0_RDREQ_128B_sum", 128))
                      if k in rd[n] and rd[n][k])
        launches = round(per_pass)
        bytes_launch = f_kib * 1024.0 * factor + w_kib * 1024.0
        res["kernels"][k] = {"launches_per_pass": launches, "dispatches_seen": len(v), "FETCH_SIZE_KiB": f_kib,
                             "WRITE_SIZE_KiB": w_kib, "bytes_per_launch": round(bytes_launch),
                             "bytes_per_launch_by_request_size": round(by_size + w_kib * 1024.0)}
        total += launches * bytes_launch
        total_by_size += launches * (by_size + w_kib * 1024.0)
    res["passes_seen"] = passes
    res["hbm_bytes_per_launch"] = round(total)              # (the key bench.py reads: one "launch" of a solver line = one pass)
    res["hbm_bytes_per_pass_by_request_size"] = round(total_by_size)
print(json.dumps(res, indent=1))
(out / f"traffic_{workload}_{solver}_{tag}.json").write_text(json.dumps(res, indent=1))
