"""The two things the merge kernel can fold into a product launch, against the oracle:

* the dot epilogue (``cask_hip_spmv_dot_device``): y = A x and w.y from one pass -- the
  "Ap = A p ; p.Ap" pair of the reference's CG (src/runtime/SparseLinearSolvers.hpp:206-208);
* halo sources (``cask_hip_csr_set_halo_sources``): columns >= n_own are read by the kernel from an
  address table, which in a sharded product points into the peers' slices.  Here, on one GPU, the
  table points into a second local buffer in scrambled order, so every code path of a seam block
  (window staging, chunked tiles, plain gathers, long rows) is exercised without a second process;
  tests/test_p2p_gpu.py runs the same kernel across processes.

Tolerance: the reference's (test/test_utils.hpp:36); repeated launches must be bitwise identical."""
import numpy as np
import pytest

import oracle
from cask_amd import capi, p2p, synth
from conftest import have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a GPU")]

MERGE_POINTS = [
    dict(variant="merge"),
    dict(variant="merge", items_per_thread=4, wg_size=128, tile_width=-1),
    dict(variant="merge", items_per_thread=8, wg_size=256, tile_width=2048, index16=-1),
    dict(variant="merge", items_per_thread=2, wg_size=64, tile_width=256, xcd_remap=-1),
    dict(variant="merge", items_per_thread=16, wg_size=256, tile_width=4096),
    dict(variant="merge", items_per_thread=8, wg_size=256, tile_width=4096, index16=2),
]
IDS = ["-".join(f"{k[:3]}{v}" for k, v in dp.items()) for dp in MERGE_POINTS]


def long_row_matrix(n=3000, seed=2):
    """Short rows plus three rows long enough to be split over several workgroups."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 6, n)
    lens[[5, n // 2, n - 1]] = [n, 70_000 % n + 900, n - 3]
    rp = np.zeros(n + 1, dtype=np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens]).astype(np.int32)
    va = rng.standard_normal(ci.size)
    return n, rp, ci, va


CASES = {
    "cant": lambda: synth.small("cant"),
    "webbase": lambda: synth.small("webbase-1M"),
    "atmosmodd": lambda: synth.small("atmosmodd"),
    "long_rows": long_row_matrix,
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dp", MERGE_POINTS + [dict(variant="vector", lanes_per_row=8), dict(variant="merge_wave")],
                         ids=IDS + ["vector", "merge_wave"])
def test_product_with_dot_epilogue(case, dp):
    import torch
    n, rp, ci, va = CASES[case]()
    rng = np.random.default_rng(3)
    x, w = rng.uniform(-1, 1, n), rng.standard_normal(n)
    want_y = oracle.csr_spmv(rp, ci, va, x)
    want_dot = float(np.dot(w, want_y))
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
    xt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    yt = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    got = []
    for _ in range(2):
        out.fill_(-1.0)
        m.spmv_dot_device(xt, yt, wt, out)
        torch.cuda.synchronize()
        got.append((yt.cpu().numpy().copy(), float(out[0])))
    oracle.assert_almost_equal(got[0][0], want_y, what=f"{case} y")
    scale = float(np.abs(w * want_y).sum())
    assert abs(got[0][1] - want_dot) <= 1e-12 * max(scale, 1.0), (got[0][1], want_dot)
    assert got[0][1] == got[1][1] and np.array_equal(got[0][0], got[1][0])      # reproducible
    m.close()


def _with_halo(n, rp, ci, va, n_own, dp):
    """Treat columns >= n_own as halo columns served from a scrambled side buffer; returns (y, want)."""
    import torch
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    n_halo = n - n_own
    perm = rng.permutation(n_halo)                                # where halo column j lives in the side buffer
    side = torch.zeros(max(n_halo, 1), dtype=torch.float64, device="cuda")
    side[torch.from_numpy(perm).cuda()] = torch.from_numpy(x[n_own:]).cuda()
    addr = torch.from_numpy((side.data_ptr() + 8 * perm).astype(np.int64)).cuda()
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
    x_own = torch.from_numpy(x[:n_own].copy()).cuda()             # the kernel must not read past n_own
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    m.set_halo_sources(n_own, addr)
    m.spmv_device(x_own, y)
    torch.cuda.synchronize()
    got = y.cpu().numpy().copy()
    # a design-point change keeps the halo; a second launch gives the same bits
    m.set_params(capi.make_params(**dp))
    y.zero_()
    m.spmv_device(x_own, y)
    torch.cuda.synchronize()
    assert np.array_equal(got, y.cpu().numpy())
    # back to the plain layout: the same handle reads all of x again
    m.set_halo_sources(0, None)
    m.spmv_device(torch.from_numpy(x).cuda(), y)
    torch.cuda.synchronize()
    oracle.assert_almost_equal(y.cpu().numpy(), want, what="plain layout restored")
    m.close()
    return got, want


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dp", MERGE_POINTS, ids=IDS)
def test_halo_columns_read_by_the_kernel(case, dp):
    n, rp, ci, va = CASES[case]()
    for n_own in (n - n // 7, n // 2):
        got, want = _with_halo(n, rp, ci, va, n_own, dp)
        oracle.assert_almost_equal(got, want, what=f"{case} n_own={n_own}")


def test_halo_needs_the_merge_variant():
    import torch
    n, rp, ci, va = synth.small("cant")
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="vector", lanes_per_row=8))
    addr = torch.zeros(16, dtype=torch.int64, device="cuda")
    with pytest.raises((capi.CaskHipError, ValueError)):
        m.set_halo_sources(n - 16, addr)
    m.set_params(capi.make_params(variant="merge"))
    m.set_halo_sources(n - 16, addr)
    with pytest.raises((capi.CaskHipError, ValueError)):
        m.set_params(capi.make_params(variant="vector", lanes_per_row=8))
    m.close()


def test_an_auto_handle_that_resolved_to_scan_accepts_halo_sources():
    """AUTO resolves to the SCAN kernel for short, heavily skewed rows (webbase-like); such a handle must
    still take halo sources -- with them AUTO resolves to MERGE -- while an explicitly requested SCAN must not."""
    n, rp, ci, va, _ = synth.load_or_make("webbase-1M")
    got, want = _with_halo(n, rp, ci, va, n - n // 5, {})         # AUTO
    oracle.assert_almost_equal(got, want, what="webbase-like, AUTO with halo sources")
    import torch
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    assert m.params.as_dict()["variant"] == "scan"
    m.close()
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="scan"))
    with pytest.raises((capi.CaskHipError, ValueError)):
        m.set_halo_sources(n - 16, torch.zeros(16, dtype=torch.int64, device="cuda"))
    m.close()


def test_p2p_symbols_exported():
    lib = capi.load()
    for sym in p2p.P2P_SYMBOLS:
        assert hasattr(lib, sym), sym


def test_dot_epilogue_inside_a_graph():
    """The fused product+dot allocates nothing at launch time, so it can be captured."""
    import torch
    n, rp, ci, va = synth.small("G3_circuit")
    rng = np.random.default_rng(4)
    x, w = rng.uniform(-1, 1, n), rng.standard_normal(n)
    want_y = oracle.csr_spmv(rp, ci, va, x)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="merge"))
    xt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    yt = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        m.spmv_dot_device(xt, yt, wt, out)          # also warms the one-off scratch of the plain dot path
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m.spmv_dot_device(xt, yt, wt, out)
    yt.zero_()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    oracle.assert_almost_equal(yt.cpu().numpy(), want_y)
    assert abs(float(out[0]) - float(np.dot(w, want_y))) <= 1e-12 * float(np.abs(w * want_y).sum())
    m.close()
