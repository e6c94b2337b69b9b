"""Non-finite operands stay where the reference puts them.

`CsrMatrix::dot` (src/runtime/SparseMatrix.hpp:255-264) touches, for row r, exactly the stored entries of row r: a NaN
or an infinity in x[c] reaches the rows that store an entry in column c and no other.  Every device kernel here pads,
clamps or masks somewhere (clamped 16-byte pair loads, foreign elements of an odd block start, padded x windows, window
slots nobody references, masked DPP steps), and a `0 * x` in any of those places would turn a neighbour's row into NaN
without any finite-input test noticing.  So: poison one entry of x -- a column referenced near the diagonal, one
referenced from far away, the first and last columns, a column NOBODY references, a halo column -- with NaN, +Inf and
-Inf, and require the set of non-finite outputs (and which of them are NaN / +Inf / -Inf) to equal the oracle's, all
other rows within the reference's tolerance.  The same for the reference's stream format decoded on the GPU (against the
oracle's decoder of that stream) and for the triangular solves (`mkl_dcsrtrsv` as `unittrsolve` calls it,
MklLayer.hpp:29-85: a bad right-hand-side entry reaches its true dependents only), in every schedule."""
import numpy as np
import pytest

import oracle
from cask_amd import capi, synth
from test_spmv_gpu import DESIGN_POINTS, DP_IDS

pytestmark = pytest.mark.gpu

BAD = (np.nan, np.inf, -np.inf)


def same_class(got, want, what):
    """Non-finite entries: the same positions, the same kind; finite ones: the reference's tolerance."""
    g_nan, w_nan = np.isnan(got), np.isnan(want)
    assert np.array_equal(g_nan, w_nan), (what, "NaN rows differ", np.flatnonzero(g_nan != w_nan)[:10])
    g_inf, w_inf = np.isinf(got), np.isinf(want)
    assert np.array_equal(g_inf, w_inf), (what, "Inf rows differ", np.flatnonzero(g_inf != w_inf)[:10])
    assert np.array_equal(np.sign(got[g_inf]), np.sign(want[w_inf])), (what, "sign of an infinity")
    fin = ~(w_nan | w_inf)
    oracle.assert_almost_equal(got[fin], want[fin], what=what)


def poison_columns(n_rows, n_cols, rp, ci):
    """Columns worth poisoning: referenced by a row next to it, by a row far from it, the first and last referenced
    ones, and one nobody references (there always is one: the callers append unreferenced columns)."""
    rows = np.repeat(np.arange(n_rows), np.diff(rp))
    dist = np.abs(ci.astype(np.int64) - rows)
    picks = {}
    if ci.size:
        picks["near"] = int(ci[np.argmin(dist)])
        picks["far"] = int(ci[np.argmax(dist)])
        picks["first"] = int(ci.min())
        picks["last"] = int(ci.max())
        counts = np.bincount(ci, minlength=n_cols)
        picks["hub"] = int(np.argmax(counts))                   # the most referenced column
    picks["unreferenced"] = n_cols - 2
    return picks


_FAMILIES = list(synth.GENERATORS)
# every design point on two of the five families (which two rotates with the point: every family meets every kernel
# family several times; all 5 x 32 cost the suite 10 s for nothing a third pairing would find)
_CASES = [(DESIGN_POINTS[i], _FAMILIES[(i + k) % len(_FAMILIES)]) for i in range(len(DESIGN_POINTS)) for k in (0, 2)]


@pytest.mark.parametrize("dp,name", _CASES, ids=[f"{DP_IDS[DESIGN_POINTS.index(dp)]}-{name}" for dp, name in _CASES])
def test_spmv_poisoned_column_reaches_its_rows_only(dp, name):
    n, rp, ci, va = synth.small(name)
    n_cols = n + 3                                              # three columns nobody references, behind the last real one
    rng = np.random.default_rng(21)
    x0 = rng.uniform(-1, 1, n_cols)
    m = capi.CsrMatrix.from_host(n, n_cols, rp, ci, va, capi.make_params(**dp))
    try:
        oracle.assert_almost_equal(m.spmv(x0), oracle.csr_spmv(rp, ci, va, x0), what=f"{name} {dp} finite")
        for where, c in poison_columns(n, n_cols, rp, ci).items():
            for bad in BAD:
                x = x0.copy()
                x[c] = bad
                want = oracle.csr_spmv(rp, ci, va, x)
                if where == "unreferenced":
                    assert np.all(np.isfinite(want))
                same_class(m.spmv(x), want, f"{name} {dp} x[{c}] ({where}) = {bad}")
    finally:
        m.close()


@pytest.mark.parametrize("key", ["matrices/test_long_row", "matrices/test_some_empty_rows", "matrices/test_tols90",
                                 "matrices/bfwb62", "matrices/test_wa"])
def test_spmv_poisoned_column_on_reference_fixtures(key):
    from oracle import mmio
    from conftest import golden_matrix_files
    m = mmio.read_matrix(dict(golden_matrix_files())[key])
    x0 = mmio.test_vector(m.m) + 1.0
    for dp in DESIGN_POINTS[::3]:                                # every family, a third of the points
        h = capi.CsrMatrix.from_host(m.n, m.m, m.row_ptr, m.col_ind, m.values, capi.make_params(**dp))
        try:
            cols = sorted(set(poison_columns(m.n, m.m, m.row_ptr, m.col_ind).values()) | {0, m.m - 1})
            for c in cols:
                for bad in BAD:
                    x = x0.copy()
                    x[c] = bad
                    same_class(h.spmv(x), oracle.csr_spmv(m.row_ptr, m.col_ind, m.values, x), f"{key} {dp} x[{c}] = {bad}")
        finally:
            h.close()


@pytest.mark.parametrize("name", ["cant", "webbase-1M", "atmosmodd"])
def test_poisoned_halo_column(name):
    """Row-sharded product with in-kernel halo reads: a bad value behind the halo address table."""
    import torch
    n, rp, ci, va = synth.small(name)
    n_own = n - n // 5
    rng = np.random.default_rng(22)
    x0 = rng.uniform(-1, 1, n)
    referenced = np.unique(ci[ci >= n_own])
    assert referenced.size
    unref = np.setdiff1d(np.arange(n_own, n), referenced)
    targets = [int(referenced[0]), int(referenced[-1])] + ([int(unref[0])] if unref.size else [])
    for dp in (dict(variant="merge"), dict(variant="merge", tile_width=-1), dict(variant="merge", items_per_thread=4, wg_size=128, tile_width=512)):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
        side = torch.zeros(n - n_own, dtype=torch.float64, device="cuda")
        addr = torch.from_numpy((side.data_ptr() + 8 * np.arange(n - n_own)).astype(np.int64)).cuda()
        m.set_halo_sources(n_own, addr)
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        try:
            for c in targets:
                for bad in BAD:
                    x = x0.copy()
                    x[c] = bad
                    side.copy_(torch.from_numpy(x[n_own:]))
                    m.spmv_device(torch.from_numpy(x[:n_own].copy()).cuda(), y)
                    torch.cuda.synchronize()
                    same_class(y.cpu().numpy(), oracle.csr_spmv(rp, ci, va, x), f"{name} {dp} halo x[{c}] = {bad}")
        finally:
            m.close()


def test_dfe_stream_poisoned_column():
    """The reference's own stream format pads every row to the input width with (0.0, column 0 of the block) entries
    (Spmv.cpp:42-107): its device multiplies them like any other, so what a bad x entry reaches is a property of the
    STREAM.  The GPU decoder must agree with the oracle's decoder of the same stream, entry for entry."""
    from oracle import dfe_format, mmio
    from conftest import golden_matrix_files
    from test_dfe_compat_gpu import Cfg, make_triple
    pipes, ctrls, cache, width = 2, 2, 32, 4
    triple, lib = make_triple(Cfg(pipes, ctrls, cache, width))
    m = mmio.read_matrix(dict(golden_matrix_files())["matrices/test_cage6"])
    parts = dfe_format.preprocess(m.n, m.m, m.row_ptr, m.col_ind, m.values, pipes, cache, width)
    x0 = mmio.test_vector(m.m) + 1.0
    for c in (0, 1, 31, 32, 50, m.m - 1):
        for bad in BAD:
            x = x0.copy()
            x[c] = bad
            got = dfe_format.spmv_through_triple(triple, m.n, parts, x, pipes, ctrls, cache)
            xp = np.concatenate([x, np.zeros((-x.size) % cache)])
            want = np.concatenate([oracle.partition_decode_spmv(p["n"], p["n_blocks"], cache, width, False, p["colptr"],
                                                                 np.frombuffer(p["records"].tobytes(), dtype=np.uint8), xp)
                                   for p in parts])[: m.n]
            same_class(got[: m.n], want, f"dfe stream x[{c}] = {bad}")
    lib.cask_hip_dfe_reset()


def _band_factor(n, per_row, band, seed):
    """A lower-triangular band: rows of up to `per_row` entries within `band` of the diagonal (long rows: lane groups)."""
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        k = min(per_row, i)
        cols = np.unique(i - 1 - rng.integers(0, max(min(band, i), 1), size=k)) if k else np.empty(0, np.int64)
        rows.append(np.concatenate([cols[(cols >= 0) & (cols < i)], [i]]))
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
    ci = np.concatenate(rows).astype(np.int32)
    va = rng.uniform(-0.2, 0.2, ci.size)
    va[rp[1:] - 1] = rng.uniform(2.0, 3.0, n)
    return n, rp, ci, va


@pytest.mark.parametrize("mode", ["default", "levels", "packed", "walk2", "lanes", "lanes4", "lanes16"])
def test_trsolve_poisoned_rhs_reaches_its_dependents_only(mode, monkeypatch):
    """`unittrsolve` (MklLayer.hpp:29-85) by forward / backward substitution: a bad right-hand-side entry makes row i and
    everything that depends on it non-finite, nothing else -- in every schedule (the lane-group walk masks the DPP steps
    of narrower groups; a `0 * Inf` there would poison a slab neighbour that depends on nothing bad)."""
    monkeypatch.delenv("CASK_HIP_TRSV", raising=False)
    monkeypatch.delenv("CASK_HIP_TRSV_LANES_E", raising=False)
    if mode.startswith("lanes") and mode != "lanes":
        monkeypatch.setenv("CASK_HIP_TRSV", "lanes")
        monkeypatch.setenv("CASK_HIP_TRSV_LANES_E", mode[5:])
    elif mode != "default":
        monkeypatch.setenv("CASK_HIP_TRSV", mode)
    import scipy.sparse as sp
    rng = np.random.default_rng(23)
    cases = {
        "band of long rows": _band_factor(3000, 24, 60, 1),     # levels of a few rows x ~20 entries: lane groups of 2-8
        "mixed widths": _band_factor(2500, 70, 400, 2),         # rows of 1 .. ~70 entries: groups of 1 .. 16 in one slab
        "short rows": _band_factor(6000, 3, 5000, 3),           # walk2 / packed territory, sources beyond the ring
    }
    for what, (n, rp, ci, va) in cases.items():
        b0 = rng.standard_normal(n)
        at = sp.csr_matrix((va, ci, rp), shape=(n, n)).T.tocsr()
        at.sort_indices()
        upper = (at.indptr.astype(np.int32), at.indices.astype(np.int32), at.data.copy())
        for lower, (trp, tci, tva) in ((True, (rp, ci, va)), (False, upper)):
            oracle.assert_almost_equal(capi.trsolve(n, trp, tci, tva, b0, lower=lower),
                                       oracle.trsolve(trp, tci, tva, b0, lower=lower), what=f"{what} {mode} finite")
            # a row nobody depends on late in the order, one early (many dependents), one in the middle
            for i in ((n - 1, 7, n // 2) if lower else (0, n - 8, n // 2)):
                for bad in BAD:
                    b = b0.copy()
                    b[i] = bad
                    want = oracle.trsolve(trp, tci, tva, b, lower=lower)
                    assert np.isfinite(want).sum() > 0 or i not in (n - 1, 0)
                    same_class(capi.trsolve(n, trp, tci, tva, b, lower=lower), want,
                               f"{what} {mode} {'lower' if lower else 'upper'} b[{i}] = {bad}")
