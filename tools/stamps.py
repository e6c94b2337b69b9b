#!/usr/bin/env python3
"""Where does a merge workgroup spend its life?  Runs the diagnostic build
(build/libcask_hip_stamps.so, -DCASK_STAMPS) on the cant workload and prints, per phase, the
distribution of durations plus the launch's timeline.  Stamps are s_memrealtime ticks (10 ns).
Phases: 0 entry, 1 loads issued (descriptor arrived), 2 window+offsets in LDS (barrier 1),
3 stream+gathers landed, 4 products in LDS (barrier 2), 5 rows reduced and stored."""
import ctypes, sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from cask_amd import capi, synth
capi.LIB_PATH = REPO / "build" / "libcask_hip_stamps.so"
import torch

def main():
    wg, ipt = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 4
    name = sys.argv[3] if len(sys.argv) > 3 else "cant"           # any BASELINE look-alike
    tile = int(sys.argv[4]) if len(sys.argv) > 4 else 1024        # 0 = AUTO, -1 = untiled
    variant = sys.argv[5] if len(sys.argv) > 5 else "merge"       # "scan": phases 0 entry, 1 loads issued, 2 stream + gathers
    #   landed, 3 products in LDS (barrier), 4 rows stored
    n, rp, ci, va, _ = synth.load_or_make(name)
    dev = torch.device("cuda", 0)
    copies = 13 if name == "cant" else 5
    rp_t = torch.from_numpy(rp).to(dev)
    lanes = int(sys.argv[6]) if len(sys.argv) > 6 else 32         # "vector": lanes per row; phases 0 entry, 1 row bounds arrived
    #   (stream requested), 2 window parked, 3 stream landed, 4 gathers + FMAs, 5 rows stored
    prm = (capi.make_params(variant=variant, wg_size=wg, lanes_per_row=lanes, tile_width=tile) if variant == "vector" else
           capi.make_params(variant=variant, wg_size=wg, items_per_thread=ipt, tile_width=tile))
    print(name, variant, "wg", wg, "items", ipt, "tile", tile)
    mats = [capi.CsrMatrix.from_device(n, n, rp_t, torch.from_numpy(ci).to(dev), torch.from_numpy(va).to(dev), prm)
            for _ in range(copies)]
    x = torch.from_numpy(np.arange(n) * 0.25 / n).to(dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    grid = mats[0].info.grid
    buf = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
    lib = capi.load()
    for i in range(copies + 3):              # cold rotation, then stamp one launch
        mats[i % copies].spmv_device(x, y)
    torch.cuda.synchronize()
    assert lib.cask_hip_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    mats[3].spmv_device(x, y)
    torch.cuda.synchronize()
    lib.cask_hip_debug_set_stamps(None)
    t = buf.cpu().numpy().reshape(grid, 8)[:, :6].astype(np.float64) * 0.01   # us
    if variant == "scan":
        t[:, 5] = t[:, 4]
        t = t[t[:, 4] > 0]                                        # long-row pieces leave no stamps
        grid = t.shape[0]
    t0 = t[:, 0].min()
    print(f"grid {grid}  launch span {t[:, 5].max() - t0:.2f} us")
    names = ["entry->issued(desc)", "issued->bar1(window,rp)", "bar1->stream+gathers", "->bar2(products)", "->reduced"]
    for k in range(5):
        d = t[:, k + 1] - t[:, k]
        print(f"  {names[k]:26s} median {np.median(d):6.2f}  p10 {np.percentile(d,10):6.2f}  p90 {np.percentile(d,90):6.2f} us")
    life = t[:, 5] - t[:, 0]
    print(f"  workgroup lifetime         median {np.median(life):6.2f}  p10 {np.percentile(life,10):6.2f}  p90 {np.percentile(life,90):6.2f} us")
    start = np.sort(t[:, 0] - t0)
    end = np.sort(t[:, 5] - t0)
    for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
        i = min(grid - 1, int(q * grid) - 1)
        print(f"  {int(q*100):3d}% of workgroups started by {start[i]:6.2f} us, finished by {end[i]:6.2f} us")

if __name__ == "__main__":
    main()
