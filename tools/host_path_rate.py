#!/usr/bin/env python3
"""Per-call time of the host-vector entry point cask_hip_spmv (x up, product, y down; the matrix stays resident) by
the way the vectors travel (include/cask_hip.h): pageable (ABI <= 6), staged with 1 / 2 / 4 / 8 host threads, registered
(the caller's vectors declared with cask_hip_host_register) and the engine's registration cache -- the C call itself,
with x and y allocated once (a fresh numpy array per call adds ~30 us of page faults that are not the engine's).
    python3 tools/host_path_rate.py [matrix ...]          -> profiles/r06_host_entry.txt is this output"""
import ctypes
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def measure(name, mode, threads):
    """One process per arm: the thread count of the copy pool is read once."""
    code = f"""
import sys, time, ctypes, json, numpy as np
sys.path.insert(0, {str(REPO)!r})
from cask_amd import capi, synth
n, rp, ci, va, _ = synth.load_or_make({name!r})
m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
L = capi.load()
x = np.arange(n, dtype=np.float64) * 0.25 / n
y = np.zeros(n)
mode = {mode!r}
t_reg = 0.0
if mode == "registered":
    t0 = time.perf_counter(); capi.host_register(x); capi.host_register(y); t_reg = time.perf_counter() - t0
else:
    capi.host_entry_mode(mode)
px, py = x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p)
for _ in range(20):
    L.cask_hip_spmv(m._h, px, py)
best = []
for rep in range(7):
    t0 = time.perf_counter()
    for _ in range(200):
        L.cask_hip_spmv(m._h, px, py)
    best.append((time.perf_counter() - t0) / 200)
import oracle
ok = oracle.mismatches(y, oracle.csr_spmv(rp, ci, va, x))[0] == 0
print(json.dumps(dict(matrix={name!r}, n=int(n), nnz=int(ci.size), mode=mode, threads={threads}, usec=round(min(best) * 1e6, 1),
                      usec_median=round(sorted(best)[3] * 1e6, 1), register_usec=round(t_reg * 1e6, 1), right=bool(ok))))
"""
    env = dict(os.environ, CASK_HIP_HOST_THREADS=str(threads))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED", name, mode, threads, out.stderr[-400:])
        return None
    print(line[-1], flush=True)
    return json.loads(line[-1])


def main():
    for name in (sys.argv[1:] or ["cant", "G3_circuit"]):
        measure(name, "pageable", 1)
        for t in (1, 2, 4, 8):
            measure(name, "staged", t)
        measure(name, "registered", 1)
        measure(name, "register_cache", 1)


if __name__ == "__main__":
    main()
