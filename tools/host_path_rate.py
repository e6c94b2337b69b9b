#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-vector entry point cask_hip_spmv (x up, kernel, y down per call;
the matrix stays resident) -- the number DESIGN.md quotes next to the device-resident `value`."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, synth

for name in (sys.argv[1:] or ["cant"]):
    n, rp, ci, va, src = synth.load_or_make(name)
    t0 = time.perf_counter()
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    t_up = time.perf_counter() - t0
    x = np.arange(n, dtype=np.float64) * 0.25 / n
    for _ in range(5):
        m.spmv(x)
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        m.spmv(x)
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: upload+plan {t_up*1e3:.1f} ms once; host-vector spmv {dt*1e6:.1f} us/call "
          f"= {2*ci.size/dt/1e9:.1f} GFLOP/s PCIe-inclusive (x {8*n/1e6:.2f} MB up, y {8*n/1e6:.2f} MB down)")
    m.close()
