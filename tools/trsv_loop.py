#!/usr/bin/env python3
"""Development: keep the ILU(0) application running for a while (argument: seconds) so that clocks can be sampled
from another process (rocm-smi --showclocks)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, synth
n, rp, ci, va, _ = synth.load_or_make("G3_circuit")
pc = capi.Preconditioner("ilu0_unit", n, rp, ci, va)
r = torch.from_numpy(np.random.default_rng(1).standard_normal(n)).cuda()
z = torch.zeros_like(r)
print("running", flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(10):
        pc.apply_device(r, z)
    torch.cuda.synchronize()
print("done", flush=True)
