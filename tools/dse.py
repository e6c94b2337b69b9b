#!/usr/bin/env python3
"""DSE driver: the GPU counterpart of `build/main <bench-path> <dse-params.json>` +
`cask.py runDse` in the reference (src/main.cpp:119-207, src/frontend/cask.py:90-122).

    python tools/dse.py [--params params.json] [--out dse_out.json] <matrix> [<matrix> ...]

<matrix> is a MatrixMarket file, a directory of them (like test/test-benchmark), or one of the
synthetic BASELINE names (cant, G3_circuit, webbase-1M, atmosmodd).  params.json keeps the
reference's range schema ({"dse_params": {"<name>": {"start","stop","step"}}}) with the GPU
parameter names lanes_per_row / tile_width / wg_size / items_per_thread; a key may also hold an
explicit {"values": [...]} list.  Every point is measured cold (rotating copies) and the winner
per matrix is written to dse_out.json with its measured GFLOP/s and % of HBM peak.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def load_ranges(path):
    if not path:
        return None
    spec = json.loads(Path(path).read_text())["dse_params"]
    out = {}
    for key, r in spec.items():
        if "values" in r:
            out[key] = list(r["values"])
        else:
            vals, v = [], r["start"]
            while v <= r["stop"]:
                vals.append(v)
                v += r["step"]
            out[key] = vals
    return out


def load_matrix(arg):
    from cask_amd import synth
    if arg in synth.GENERATORS:
        n, rp, ci, va, src = synth.load_or_make(arg)
        return arg, n, n, rp, ci, va
    import scipy.io
    import scipy.sparse as sp
    a = sp.csr_matrix(scipy.io.mmread(arg))
    a.sum_duplicates()
    a.sort_indices()
    return arg, a.shape[0], a.shape[1], a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matrices", nargs="+")
    ap.add_argument("--params", default=None)
    ap.add_argument("--out", default="dse_out.json")
    ap.add_argument("--steps", type=int, default=0, help="launches per timed graph (0 = the engine default)")
    args = ap.parse_args()
    import torch
    from cask_amd import capi, dse

    paths = []
    for m in args.matrices:
        p = Path(m)
        paths += sorted(str(f) for f in p.glob("*.mtx")) if p.is_dir() else [m]
    points = dse.design_points(load_ranges(args.params))
    entries, t0 = [], time.perf_counter()
    dev = torch.device("cuda", 0)
    for path in paths:
        name, n_rows, n_cols, rp, ci, va = load_matrix(path)
        rp_t = torch.from_numpy(rp).to(dev)
        mats = [capi.CsrMatrix.from_device(n_rows, n_cols, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev))]
        rows, best, took = dse.explore(mats, points=points, steps=args.steps)     # cask_hip_tune: cold copies made inside
        print(f"{Path(name).name}: best {best['variant']} {best['usec']} us cold ({best['usec_warm']} warm)  "
              f"{best['gflops']} GFLOP/s  {best['pct_hbm_peak']}% of HBM peak  ({len(rows)} points, {took:.1f} s)")
        entries.append({"matrix": name, "best": best, "points": len(rows), "rows": rows})
        for m in mats:
            m.close()
    dse.write_dse_out(args.out, entries, time.perf_counter() - t0)
    Path(args.out).with_suffix(".points.json").write_text(json.dumps(
        [{"matrix": e["matrix"], "rows": e["rows"]} for e in entries], indent=1))


if __name__ == "__main__":
    main()
