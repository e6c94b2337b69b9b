#!/usr/bin/env python3
"""VERDICT r3 item 5b, feasibility before any kernel is written: "launch 1 = near-only product (far pre-gather riding along as
independent workgroups), launch 2 = streaming far accumulate y[row] += v * farx[k] in row order".

Both launches can be timed with the engine as it is:
  T1  the product of A_near (the webbase-like matrix without its far nonzeros: columns further than `margin` from the row)
  T2  the product of A_far with its columns REPLACED by 0, 1, 2, ... in row order -- exactly the access pattern of a streaming
      accumulate that reads farx sequentially (+ the y it would have to read back: not included, so T2 is optimistic)
  T0  the whole matrix (what the engine does today)
The split can only pay if T1 + T2 (+ 1.45 us for the second launch boundary, with the pre-gather hidden completely) < T0.
    python tools/webbase_split_estimate.py [webbase-1M|webbase2] [margin]
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, dse, synth  # noqa: E402


def cold_time(n, rp, ci, va, label):
    import torch
    matrix_bytes = 12 * ci.size + 4 * (n + 1)
    copies = max(2, -(-2 * (256 << 20) // max(matrix_bytes, 1)) + 1)
    mats = [capi.CsrMatrix.from_host(n, n, rp, ci, va) for _ in range(min(copies, 24))]
    x = torch.arange(n, dtype=torch.float64, device="cuda") * 0.25 / n
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    rows, best, took = dse.explore(mats, x, y)
    out = {"what": label, "nnz": int(ci.size), "usec_cold_best": round(best["usec"], 2), "design": {k: best[k] for k in ("variant", "wg_size", "items_per_thread", "tile_width")}}
    for m in mats:
        m.close()
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
    margin = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    n, rp, ci, va, _ = synth.load_or_make(name)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    far = np.abs(ci.astype(np.int64) - rows) > margin

    def sub(mask, cols):
        r = rows[mask]
        nrp = np.zeros(n + 1, dtype=np.int64)
        np.add.at(nrp, r + 1, 1)
        return np.cumsum(nrp).astype(np.int32), cols.astype(np.int32), va[mask]

    res = [cold_time(n, rp, ci, va, "whole matrix (today)")]
    nrp, nci, nva = sub(~far, ci[~far])
    res.append(cold_time(n, nrp, nci, nva, f"A_near (|col - row| <= {margin})"))
    k = int(far.sum())
    frp, fci, fva = sub(far, np.arange(k) % n)                  # sequential "farx" positions
    res.append(cold_time(n, frp, fci, fva, "A_far with sequential columns (streaming accumulate, optimistic)"))
    t0, t1, t2 = (r["usec_cold_best"] for r in res)
    print(json.dumps({"matrix": name, "margin": margin, "far_nnz": k, "runs": res,
                      "estimate_split_usec": round(t1 + t2 + 1.45, 2), "today_usec": t0}, indent=1))


if __name__ == "__main__":
    main()
