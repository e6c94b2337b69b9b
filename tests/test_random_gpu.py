"""Randomised parity sweep: random sparsity patterns (empty rows, long rows, narrow and wide column
windows, odd and even nonzero counts) against random design points, product and product+dot, compared
with the oracle under the reference's tolerance (test/test_utils.hpp:36).  Seeds are fixed; the point
is breadth over the planner's corner cases (block boundaries on odd nonzeros, last half pair, tiles that
fit / do not fit, packed and 16-bit slot layouts, split rows), not randomness per run."""
import numpy as np
import pytest

import oracle
from cask_amd import capi
from conftest import have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a GPU")]


def random_csr(rng, n_rows, n_cols, kind):
    if kind == "banded":
        lens = rng.integers(0, 40, n_rows)
        half = int(rng.integers(8, 400))
    elif kind == "powerlaw":
        lens = np.minimum((rng.pareto(1.3, n_rows) * 2).astype(np.int64), n_cols)
        half = n_cols
    else:                                   # "mixed": mostly short rows, a few very long ones
        lens = rng.integers(0, 6, n_rows)
        for r in rng.choice(n_rows, size=min(3, n_rows), replace=False):
            lens[r] = int(rng.integers(n_cols // 2, n_cols + 1))
        half = int(rng.integers(50, 3000))
    rp = np.zeros(n_rows + 1, dtype=np.int64)
    cols = []
    for r in range(n_rows):
        centre = int(r * n_cols / max(n_rows, 1))
        lo, hi = max(0, centre - half), min(n_cols, centre + half + 1)
        k = int(min(lens[r], hi - lo))
        cols.append(np.sort(rng.choice(np.arange(lo, hi), size=k, replace=False)) if k else np.zeros(0, dtype=np.int64))
        rp[r + 1] = rp[r] + k
    ci = np.concatenate(cols).astype(np.int32) if rp[-1] else np.zeros(0, dtype=np.int32)
    va = rng.standard_normal(int(rp[-1]))
    return rp.astype(np.int32), ci, va


def random_design_point(rng):
    variant = rng.choice(["merge", "merge", "merge", "vector", "merge_wave", "scan", "scan", "slice", "slice"])
    if variant == "merge":
        return dict(variant="merge", items_per_thread=int(rng.choice([2, 4, 8, 8, 16])), wg_size=int(rng.choice([64, 128, 256, 512])),
                    tile_width=int(rng.choice([-1, 64, 512, 1024, 4096])), index16=int(rng.choice([-1, 1, 1, 2, 0])),
                    xcd_remap=int(rng.choice([-1, 1])), nontemporal=int(rng.choice([-1, 1])))
    if variant == "vector":
        return dict(variant="vector", lanes_per_row=int(rng.choice([1, 2, 4, 8, 16, 32, 64])), wg_size=int(rng.choice([64, 256])),
                    tile_width=int(rng.choice([-1, 1024])))
    if variant == "scan":
        return dict(variant="scan", items_per_thread=int(rng.choice([2, 4, 8, 8, 16])), wg_size=int(rng.choice([64, 128, 256, 512])),
                    tile_width=int(rng.choice([-1, 64, 1024, 4096])),
                    xcd_remap=int(rng.choice([-1, 1])), nontemporal=int(rng.choice([-1, 1])))
    if variant == "slice":
        return dict(variant="slice", lanes_per_row=int(rng.integers(1, 9)), items_per_thread=int(rng.choice([4, 8])),
                    wg_size=int(rng.choice([64, 128, 256, 512])), tile_width=int(rng.choice([-1, 64, 1024, 4096])),
                    xcd_remap=int(rng.choice([-1, 1])))
    return dict(variant="merge_wave", items_per_thread=int(rng.choice([2, 4, 8, 16])), wg_size=int(rng.choice([64, 256])))


@pytest.mark.parametrize("seed", range(24))
def test_random_matrix_random_design_point(seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    kind = ["banded", "powerlaw", "mixed"][seed % 3]
    n_rows = int(rng.integers(1, 6000))
    n_cols = int(rng.integers(max(2, n_rows // 2), 2 * n_rows + 50)) if seed % 4 else n_rows
    rp, ci, va = random_csr(rng, n_rows, n_cols, kind)
    x = rng.uniform(-1, 1, n_cols)
    w = rng.standard_normal(n_rows)
    want = oracle.csr_spmv(rp, ci, va, x)
    for _ in range(3):
        dp = random_design_point(rng)
        try:
            m = capi.CsrMatrix.from_host(n_rows, n_cols, rp, ci, va, capi.make_params(**dp))
        except ValueError as e:                      # e.g. wg_size * items_per_thread over the LDS budget
            assert "LDS" in str(e) or "wg_size" in str(e), (dp, e)
            continue
        what = f"seed {seed} {kind} {n_rows}x{n_cols} nnz {ci.size} {dp}"
        oracle.assert_almost_equal(m.spmv(x), want, what=what)
        xt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
        yt = torch.zeros(n_rows, dtype=torch.float64, device="cuda")
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        m.spmv_dot_device(xt, yt, wt, out)
        torch.cuda.synchronize()
        oracle.assert_almost_equal(yt.cpu().numpy(), want, what=what + " (with dot)")
        scale = float(np.abs(w * want).sum())
        assert abs(float(out[0]) - float(np.dot(w, want))) <= 1e-11 * max(scale, 1.0), what
        m.close()
