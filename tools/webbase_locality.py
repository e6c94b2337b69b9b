#!/usr/bin/env python3
"""Development: how much of the webbase-like product's time is the scattered columns?  Same power-law row lengths,
fraction `p` of the columns within +-1000 of the diagonal (the BASELINE look-alike has p = 0.7), rest uniform."""
import json, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, synth

def make(p, n=1_000_005, nnz_target=3_105_536, alpha=2.1, max_row=4700, seed=3):
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.zipf(alpha, size=n), max_row).astype(np.int64)
    lens = np.clip(np.rint(lens * (nnz_target / lens.sum())), 1, max_row).astype(np.int64)
    lens[rng.integers(0, n)] = max_row
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    local = rng.random(rows.size) < p
    cols = np.clip(np.where(local, rows + rng.integers(-1000, 1001, size=rows.size), rng.integers(0, n, size=rows.size)), 0, n - 1)
    return (n,) + synth._coo_to_csr(n, rows, cols, rng.random(rows.size))

for p in (1.0, 0.9, 0.7, 0.0):
    n, rp, ci, va = make(p)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
    med, mn = m.time(x, y, warmup=10, iters=100)
    print(json.dumps({"p_local": p, "nnz": int(ci.size), "usec_median": round(med, 2), "usec_min": round(mn, 2),
                      "design_point": m.params.as_dict()["variant"], "info": {k: v for k, v in m.info.__dict__.items()} if hasattr(m.info, "__dict__") else None}))
    m.close()
