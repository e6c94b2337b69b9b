#!/usr/bin/env python3
"""Cold microseconds per launch of chosen design points on the BASELINE families (development A/B).
   tools/family_times.py webbase-1M 'variant=merge,wg_size=512,items_per_thread=4,tile_width=4096,far_columns=1' ..."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from cask_amd import capi, synth, dse

def main():
    name = sys.argv[1]
    n, rp, ci, va, _ = synth.load_or_make(name)
    dev = torch.device("cuda", 0)
    copies = dse.copies_for_cold(12 * ci.size + 4 * (n + 1))
    rp_t = torch.from_numpy(rp).to(dev)
    x = torch.from_numpy(np.arange(n) * 0.25 / n).to(dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    alg = synth.algorithmic_bytes(n, n, ci.size)
    for spec in sys.argv[2:] or ["variant=merge"]:
        kw = {}
        for kv in spec.split(","):
            k, v = kv.split("=")
            kw[k] = v if k == "variant" else int(v)
        mats = []
        for _ in range(copies):
            mats.append(capi.CsrMatrix.from_device(n, n, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev),
                                                   capi.make_params(**kw)))
        us = dse.measure(mats, x, y, steps=4 * copies, reps=5)
        info = mats[0].info
        print(json.dumps({"matrix": name, "spec": spec, "usec_cold": round(us, 3), "gflops": round(2 * ci.size / us * 1e-3, 1),
                          "pct_peak": round(alg / us * 1e-3 / 80, 1), "resolved": mats[0].params.as_dict(),
                          "grid": info.grid, "lds": info.lds_bytes}), flush=True)
        for m in mats:
            m.close()

if __name__ == "__main__":
    main()
