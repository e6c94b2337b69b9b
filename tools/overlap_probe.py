#!/usr/bin/env python3
"""What the headline's fixed per-launch cost is worth: the same cold products issued on ONE stream (dependent launches: the
bench's form, every launch pays its head and its tail in full) and on S streams (independent launches of different
matrix copies with their own y: launch i + 1's head -- dispatch ramp, descriptor trip -- overlaps launch i's tail -- row
sums, y stores).  NOT a bench line: a step of BASELINE's metric is one dependent product; this prices the overlap a
caller with independent products (several matrices, several right-hand sides) would get.
    python tools/overlap_probe.py [workload] [launches per stream] [streams ...]"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cant"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    stream_counts = [int(a) for a in sys.argv[3:]] or [1, 2, 3, 4]
    n, rp, ci, va, _ = synth.load_or_make(name)
    matrix_bytes = 12 * ci.size + 4 * (n + 1)
    copies = max(4, -(-2 * (256 << 20) // matrix_bytes) + 1)
    mats = [capi.CsrMatrix.from_host(n, n, rp, ci, va) for _ in range(copies)]
    x = torch.from_numpy(np.arange(n, dtype=np.float64) * 0.25 / n).cuda()
    out = {"workload": name, "copies": copies, "launches_per_stream": k, "design_point": mats[0].params.as_dict(), "usec_per_product": {}}
    for s_count in stream_counts:
        # ONE graph with s_count parallel chains of k kernel nodes (the host's launch rate must not be what is measured)
        ys = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(s_count)]
        side = [torch.cuda.Stream() for _ in range(s_count - 1)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cur = torch.cuda.current_stream()
            chains = [cur] + side
            for s in side:
                s.wait_stream(cur)
            for i in range(k):                                # chain j's launch i is copy (i*S + j) mod copies
                for j, s in enumerate(chains):
                    mats[(i * s_count + j) % copies].spmv_device(x, ys[j], stream=s)
            for s in side:
                cur.wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (k * s_count)
            best = us if best is None else min(best, us)
        out["usec_per_product"][str(s_count)] = round(best, 3)
        del g
    print(json.dumps(out))
    for m in mats:
        m.close()


if __name__ == "__main__":
    main()
