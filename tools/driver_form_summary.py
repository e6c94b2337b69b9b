#!/usr/bin/env python3
"""Fold the driver-form runs of every lease (gpurun_out/driver_form_<lease>.jsonl, written by tools/driver_form_runs.sh: the
exact command `python3 bench.py --gpus 1 --steps 20 --warmup 5`, several times in a row per lease) into
profiles/r04_driver_form_runs.json."""
import glob
import json
import statistics
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
runs, leases = [], {}
for f in sorted(glob.glob(str(root / "gpurun_out" / "driver_form_*.jsonl"))):
    lease = Path(f).stem.replace("driver_form_", "")
    for i, line in enumerate(open(f)):
        line = line.strip()
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        run = {"lease": lease, "run": i + 1, "usec": round(r["ms_per_step"] * 1e3, 3), "gflops": r["value"],
               "frac": r["roofline"]["frac"], "windows": r.get("windows"),
               "usec_min": round(r["ms_per_step_min"] * 1e3, 3), "usec_p10": round(r["ms_per_step_p10"] * 1e3, 3),
               "usec_p90": round(r["ms_per_step_p90"] * 1e3, 3), "usec_max": round(r["ms_per_step_max"] * 1e3, 3),
               "usec_first_window": round(r["ms_per_step_first_window"] * 1e3, 3),
               "clocks_mhz_during": r["gpu_clocks_mhz"]["during_windows"], "launch": r["config"]["launch"],
               "untimed_preroll_replays": r["config"].get("untimed_preroll_replays"),
               "preroll": "300 ms (final form)" if (r["config"].get("untimed_preroll_replays") or 0) > 800 else "40 ms (rounds 1-3 form)",
               "design_point": {k: r["config"]["design_point"][k] for k in ("variant", "wg_size", "items_per_thread", "tile_width")},
               "other_workloads_usec": {o["workload"][:40]: o.get("usec") for o in r["config"].get("other_workloads", [])}}
        runs.append(run)
        leases.setdefault(lease, []).append(run["usec"])
final = [r for r in runs if r["preroll"].startswith("300")]
us = [r["usec"] for r in runs]
doc = {"command": "python3 bench.py --gpus 1 --steps 20 --warmup 5", "runs": len(runs), "leases": len(leases),
       "usec_median_of_runs": round(statistics.median(us), 3), "usec_min": min(us), "usec_max": max(us),
       "spread_pct_all_runs": round(100.0 * (max(us) - min(us)) / statistics.median(us), 2),
       "per_lease": {k: {"runs": len(v), "usec": v, "spread_pct": round(100.0 * (max(v) - min(v)) / statistics.median(v), 2)}
                     for k, v in leases.items()},
       "frac_min": min(r["frac"] for r in runs), "frac_max": max(r["frac"] for r in runs),
       "final_form_300ms_preroll": ({"runs": len(final), "leases": len({r["lease"] for r in final}),
                                     "usec_median_of_runs": round(statistics.median([r["usec"] for r in final]), 3),
                                     "usec_min": min(r["usec"] for r in final), "usec_max": max(r["usec"] for r in final),
                                     "spread_pct": round(100.0 * (max(r["usec"] for r in final) - min(r["usec"] for r in final)) /
                                                         statistics.median([r["usec"] for r in final]), 2),
                                     "frac_min": min(r["frac"] for r in final), "frac_max": max(r["frac"] for r in final)}
                                    if final else None),
       "driver_records": {"BENCH_r02": {"usec": 8.676, "gflops": 924.6, "frac": 0.7114, "timed": "ONE 20-step window"},
                          "BENCH_r03": {"usec": 9.756, "gflops": 822.2, "frac": 0.6327, "timed": "ONE 20-step window"}},
       "detail": runs}
out = root / "profiles" / "r04_driver_form_runs.json"
out.write_text(json.dumps(doc, indent=1))
print(json.dumps({k: doc[k] for k in doc if k != "detail"}, indent=1))
