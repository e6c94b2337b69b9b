#!/usr/bin/env python3
"""What the x tiles of a MERGE plan fetch (CPU, numpy): plan::build_chunk_tiles' rule restated -- blocks of <= cap nonzeros
snapped to row ends, per block the chunks of `gran` columns that hold its columns -- and, per XCD-eighth of the blocks, the
DISTINCT 128-byte lines of x: what a perfect L2 would fetch.  profiles/r06_line_tiles.txt.
    python tools/chunk_model.py G3_circuit atmosmodd"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import synth
def model(name, cap=2048, gran=64, gap=32):
    m = synth.load_or_make(name)
    n, rp, ci = m[0], np.asarray(m[1]), np.asarray(m[2])
    nnz = len(ci)
    # blocks: consecutive pieces of <= cap nnz snapped to row ends (approximation of build_merge_blocks)
    row_of_cut = []
    starts = [0]
    r = 0
    while rp[r] < nnz:
        target = rp[r] + cap
        r2 = int(np.searchsorted(rp, target, side="right") - 1)
        if r2 <= r: r2 = r + 1
        r2 = min(r2, n)
        starts.append(r2); r = r2
        if r >= n: break
    nb = len(starts) - 1
    tot_lines = 0; out_lines = 0
    per_xcd = [set() for _ in range(8)]
    chunks_hist = []
    for b in range(nb):
        k0, k1 = rp[starts[b]], rp[starts[b+1]]
        u = np.unique(ci[k0:k1])
        # runs with gap
        brk = np.nonzero(np.diff(u) > gap)[0]
        i0 = np.concatenate([[0], brk + 1]); i1 = np.concatenate([brk, [len(u) - 1]])
        lines = []
        nch = 0
        for a, z in zip(u[i0], u[i1]):
            s = a & ~1 if gran == 64 else a & ~(gran - 1)
            k = (z - s) // gran + 1
            nch += k
            # lines of 128 B = 16 doubles
            l0 = (s) // 16; l1 = (s + k * gran - 1) // 16
            lines.append(np.arange(l0, l1 + 1))
        lines = np.unique(np.concatenate(lines))
        tot_lines += len(lines)
        chunks_hist.append(nch * gran)
        per_xcd[b * 8 // nb].update(lines.tolist())
    uniq = sum(len(s) for s in per_xcd)
    print(f"{name} gran {gran}: blocks {nb}  window slots mean {np.mean(chunks_hist):.0f} max {np.max(chunks_hist)}  "
          f"requested {tot_lines*128/1e6:.1f} MB  unique per XCD {uniq*128/1e6:.1f} MB  x itself {n*8/1e6:.1f} MB")
for name in sys.argv[1:]:
    for g in (64, 16):
        model(name, gran=g)
