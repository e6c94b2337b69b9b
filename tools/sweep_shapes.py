#!/usr/bin/env python3
"""A wider shape sweep of the merge family than the DSE's default ranges (workgroups of 128 / 1024 threads, 2 / 16 items
per thread) on one matrix, cold -- are the default ranges leaving anything on the table?  tools/sweep_shapes.py [matrix]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cask_amd import capi, dse, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cant"
n, rp, ci, va, _ = synth.load_or_make(name)
dev = torch.device("cuda", 0)
rp_t = torch.from_numpy(rp).to(dev)
m = capi.CsrMatrix.from_device(n, n, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev))
pts = [dict(variant="merge", items_per_thread=i, tile_width=t, wg_size=w) for i in (2, 4, 8, 16) for w in (128, 256, 512, 1024)
       for t in (1024, 2048)]
rows, best, took = dse.explore([m], points=pts)
for r in sorted((r for r in rows if r.get("valid")), key=lambda r: r["usec"]):
    print(f'w{r["wg_size"]:<5d} i{r["items_per_thread"]:<3d} t{r["tile_width"]:<5d} {r["usec"]:8.3f} us cold  {r["usec_warm"]:8.3f} warm')
print(json.dumps({"matrix": name, "best": {k: best[k] for k in ("wg_size", "items_per_thread", "tile_width", "usec")}, "seconds": round(took, 1)}))
