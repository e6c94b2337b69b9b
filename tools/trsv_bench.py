#!/usr/bin/env python3
"""Device time of one ILU(0) application (two triangular solves) on a BASELINE look-alike, per schedule:
    python tools/trsv_bench.py [matrix] [kind]        (CASK_HIP_TRSV=levels|packed|walk2|lanes forces a schedule)
Prints one JSON line: milliseconds per application (best of 5), dependency levels, launches."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "G3_circuit"
    kind = sys.argv[2] if len(sys.argv) > 2 else "ilu0_unit"
    n, rp, ci, va, _ = synth.load_or_make(name)
    pc = capi.Preconditioner(kind, n, rp, ci, va)
    r = torch.from_numpy(np.random.default_rng(1).standard_normal(n)).cuda()
    z = torch.zeros_like(r)
    pc.apply_device(r, z)
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pc.apply_device(r, z)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    print(json.dumps({"matrix": name, "preconditioner": kind, "schedule": os.environ.get("CASK_HIP_TRSV", "default"),
                      "ms_per_apply": round(min(times), 3), "ms_all": [round(t, 3) for t in times],
                      "checksum": float(z.sum()), **pc.info()}))
    pc.close()


if __name__ == "__main__":
    main()
