#!/usr/bin/env python3
"""Does the headline's step time drift with SUSTAINED load?  One process, the bench's matrix and rotation, back-to-back
blocks of 101 windows x 20 steps (17 ms each) for ~12 s; prints the median step of every block against the wall clock,
with the GPU clocks.  tools/drift_probe.py [seconds] [idle_seconds_between_phases]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from cask_amd import capi, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n, rp, ci, va = synth.cant_like()
dev = torch.device("cuda", 0)
prm = capi.make_params(variant="merge", lanes_per_row=16, tile_width=1024, items_per_thread=8, wg_size=256)
rp_t = torch.from_numpy(rp).to(dev)
mats = [capi.CsrMatrix.from_device(n, n, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev), prm) for _ in range(13)]
x = torch.arange(n, dtype=torch.float64, device=dev) * 0.25 / n
y = torch.zeros(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for phase in ("cold start", f"after {idle:.0f} s idle"):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        u = capi.spmv_windows_device(mats, x, y, 20, 101)
        if k < 10 or k % 25 == 0:
            print(f"{phase:18s} t = {time.perf_counter() - t0:6.2f} s   median {np.median(u) / 20:.3f}  p10 {np.percentile(u, 10) / 20:.3f}  "
                  f"p90 {np.percentile(u, 90) / 20:.3f} us/step   {bench.gpu_clocks(0)}", flush=True)
        k += 1
    time.sleep(idle)
