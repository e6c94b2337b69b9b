#!/usr/bin/env python3
"""How steady is one graph replay?  Captures K cold SpMV steps (rotating copies, as bench.py does)
and prints the per-step time of each of R back-to-back replays, with optional idle gaps between them.
Development tool: explains run-to-run spread of bench.py's single timed replay."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--replays", type=int, default=12)
    ap.add_argument("--gap-ms", type=float, default=0.0)
    args = ap.parse_args()
    import torch
    from cask_amd import capi, synth
    n, rp, ci, va, _ = synth.load_or_make("cant")
    dev = torch.device("cuda", 0)
    copies = 13
    rp_t = torch.from_numpy(rp).to(dev)
    mats = [capi.CsrMatrix.from_device(n, n, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev))
            for _ in range(copies)]
    x = torch.from_numpy(np.arange(n, dtype=np.float64) * 0.25 / n).to(dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    for i in range(20):
        mats[i % copies].spmv_device(x, y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(args.steps):
            mats[i % copies].spmv_device(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for r in range(args.replays):
        if args.gap_ms:
            time.sleep(args.gap_ms * 1e-3)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) * 1e3 / args.steps, 3))
    print(f"steps={args.steps} gap_ms={args.gap_ms} usec/step per replay: {out}")


if __name__ == "__main__":
    main()
