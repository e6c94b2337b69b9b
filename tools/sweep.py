#!/usr/bin/env python3
"""Cold-cache design-space sweep on one GPU (development tool behind the DSE).

For every design point: apply it to R rotating device copies of the matrix
(R copies exceed 2x the Infinity Cache), capture `steps` SpMVs into a HIP graph,
replay a few times and report the best per-launch time.  Prints a table sorted
by time and writes JSON under gpurun_out/ when --out is given.
"""
import argparse
import itertools
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cant")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--warm", action="store_true", help="single copy (cache-warm) instead of rotation")
    ap.add_argument("--only", default=None, help="restrict to one family: vector|merge|pipe")
    ap.add_argument("--ci-mode", default=None, choices=[None, "zero", "seq", "diag"],
                    help="ablation: replace column indices (zero: all 0; seq: contiguous run per row; diag: row index)")
    args = ap.parse_args()

    import torch
    from cask_amd import capi, synth

    n, rp, ci, va, source = synth.load_or_make(args.workload)
    nnz = int(ci.size)
    if args.ci_mode:
        rows_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
        if args.ci_mode == "zero":
            ci = np.zeros_like(ci)
        elif args.ci_mode == "diag":
            ci = rows_of.astype(np.int32)
        else:
            within = np.arange(nnz, dtype=np.int64) - np.repeat(rp[:-1].astype(np.int64), np.diff(rp))
            ci = np.minimum(rows_of + within, n - 1).astype(np.int32)
    alg = synth.algorithmic_bytes(n, n, nnz)
    mbytes = 12 * nnz + 4 * (n + 1)
    copies = 1 if args.warm else max(2, -(-2 * (256 << 20) // mbytes) + 1)
    dev = torch.device("cuda", 0)
    rp_t = torch.from_numpy(rp).to(dev)
    mats = []
    for _ in range(copies):
        mats.append(capi.CsrMatrix.from_device(n, n, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev)))
    x = torch.from_numpy(np.arange(n, dtype=np.float64) * 0.25 / n).to(dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)

    points = []
    if args.quick:
        lanes_l, wg_l, tile_l, ipt_l = [8, 16, 32], [256], [-1, 4096], [4, 8]
        flags = [(1, 1)]
    else:
        lanes_l, wg_l, tile_l, ipt_l = [2, 4, 8, 16, 32, 64], [128, 256, 512], [-1, 1024, 4096], [2, 4, 8, 16]
        flags = [(1, 1), (-1, 1), (1, -1)]
    for L, wg, tile, (xcd, nt) in itertools.product(lanes_l, wg_l, tile_l, flags):
        points.append(dict(variant="vector", lanes_per_row=L, wg_size=wg, tile_width=tile, xcd_remap=xcd, nontemporal=nt))
    for ipt, wg, tile, (xcd, nt) in itertools.product(ipt_l, wg_l, tile_l, flags):
        points.append(dict(variant="merge", items_per_thread=ipt, wg_size=wg, tile_width=tile, xcd_remap=xcd, nontemporal=nt))
        if tile > 0 and (xcd, nt) == (1, 1):
            points.append(dict(variant="merge", items_per_thread=ipt, wg_size=wg, tile_width=tile, xcd_remap=xcd,
                               nontemporal=nt, index16=-1))
    for ipt, wg, (xcd, nt) in itertools.product(ipt_l, [64, 128, 256, 512, 1024], flags):
        points.append(dict(variant="merge_wave", items_per_thread=ipt, wg_size=wg, xcd_remap=xcd, nontemporal=nt))

    if args.only == "vector":
        points = [p for p in points if p["variant"] == "vector"]
    elif args.only == "merge":
        points = [p for p in points if p["variant"] == "merge"]
    elif args.only == "pipe":
        points = [p for p in points if p["variant"] == "merge_wave"]
    rows = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for dp in points:
        try:
            prm = capi.make_params(**dp)
            for m in mats:
                m.set_params(prm)
        except ValueError as e:
            rows.append({**dp, "usec": None, "error": str(e)})
            continue
        for i in range(5):
            mats[i % copies].spmv_device(x, y)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(args.steps):
                mats[i % copies].spmv_device(x, y)
        g.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(args.reps):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / args.steps)
        info = mats[0].info
        rows.append({**dp, "usec": round(best, 3), "gbs": round(alg / best * 1e-3, 1), "gflops": round(2 * nnz / best * 1e-3, 1),
                     "grid": info.grid, "lds": info.lds_bytes})
        del g
    ok = sorted([r for r in rows if r.get("usec")], key=lambda r: r["usec"])
    print(f"# {args.workload} ({source}) n={n} nnz={nnz} alg_bytes={alg} copies={copies}")
    for r in ok[:40]:
        print(json.dumps(r))
    print("# slowest:")
    for r in ok[-5:]:
        print(json.dumps(r))
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps({"workload": args.workload, "n": n, "nnz": nnz, "algorithmic_bytes": alg,
                                              "copies": copies, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
