#!/usr/bin/env python3
"""dse_out.json with counter evidence (north_star: "the chosen variant evidenced by rocprof HBM GB/s
against the chip's peak"; the reference's writer: src/main.cpp:81-117).  Run on the GPU box:

    python tools/dse_evidence.py <tag> <out.json> <matrix> [<matrix> ...]

1. tools/dse.py (cask_hip_tune: the one DSE) -> the winner per matrix, with its algorithmic GB/s;
2. tools/pmc_point.sh at THAT design point: three separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, read requests
   by size) of bench.py forced to the winner, FETCH_SIZE corrected by the factor measured on the calibration kernel in
   the same session (tools/traffic_summary.py);
   -- and, r6, at the runner-up of the winner's family when it is within 1 % (a later run may time that one);
3. the winner's entry gains hbm_bytes_per_launch_measured, hbm_gbs_measured, pct_hbm_peak_measured (bytes the counters
   saw / the winner's cold launch time) next to measured_gbs_algorithmic."""
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
tag, out = sys.argv[1], Path(sys.argv[2])
mats = sys.argv[3:]
subprocess.run([sys.executable, str(REPO / "tools" / "dse.py"), "--out", str(out)] + mats, check=True)
doc = json.loads(out.read_text())
def profile_point(w, p):
    """Three PMC passes of bench.py forced to design point p of workload w -> (label, traffic record or None)."""
    label = f'{p["variant"]}_w{p["wg_size"]}_i{p["items_per_thread"]}_t{p["tile_width"]}_l{p["lanes_per_row"]}'.replace("-", "m")
    flags = ["--variant", p["variant"], "--wg", str(p["wg_size"]), "--tile", str(p["tile_width"])]
    if p["items_per_thread"] > 0:
        flags += ["--items", str(p["items_per_thread"])]
    if p["lanes_per_row"] > 0:
        flags += ["--lanes", str(p["lanes_per_row"])]
    subprocess.run(["bash", str(REPO / "tools" / "pmc_point.sh"), tag, w, label] + flags, check=False,
                   stdout=subprocess.DEVNULL)
    tf = REPO / "gpurun_out" / f"traffic_{w}_{label}_{tag}.json"
    return label, (json.loads(tf.read_text()) if tf.exists() else None)


for arch in doc["best_architectures"]:
    w = arch["matrices"][0]
    label, t = profile_point(w, arch["architecture_params"])
    ru = arch.get("runner_up_within_1pct")
    if ru:                                                       # within 1 % of the winner: a later run may time this one
        ru_label, ru_t = profile_point(w, ru["architecture_params"])
        ru["design_point_label"] = ru_label
        ru["traffic_file"] = f"profiles/traffic_{w}_{ru_label}.json" if ru_t else None
        ru["hbm_bytes_per_launch_measured"] = ru_t.get("hbm_bytes_per_launch") if ru_t else None
    if t is None:
        arch["hbm_bytes_per_launch_measured"] = None
        continue
    b = t.get("hbm_bytes_per_launch")
    arch["design_point_label"] = label
    arch["traffic_file"] = f"profiles/traffic_{w}_{label}.json"
    arch["hbm_bytes_per_launch_measured"] = b
    arch["hbm_bytes_per_launch_by_request_size"] = t.get("hbm_bytes_per_launch_by_request_size")
    if b and arch.get("measured_usec"):
        gbs = b / arch["measured_usec"] * 1e-3
        arch["hbm_gbs_measured"] = round(gbs, 1)
        arch["pct_hbm_peak_measured"] = round(100.0 * gbs / 8000.0, 2)
    factor = t["calibration"].get("factor")
    arch["traffic_method"] = ("rocprofv3 --pmc, three separate passes at this design point: FETCH_SIZE x calibration factor "
                              f'{factor if factor is None else round(factor, 4)} + WRITE_SIZE; cross-check: L2->memory read requests by size')
out.write_text(json.dumps(doc, indent=2))
for a in doc["best_architectures"]:
    print(a["matrices"][0], a["name"], a["measured_usec"], "us  algorithmic", a["measured_gbs_algorithmic"], "GB/s  measured",
          a.get("hbm_gbs_measured"), "GB/s", a.get("hbm_bytes_per_launch_measured"))
