"""BLAS-1, CG and BiCG on the GPU (through the C ABI) against the oracle's
restatement of the reference's pcg (src/runtime/SparseLinearSolvers.hpp:162-239)
and its known answers (test/LinearSolvers.cpp:14-52)."""
import numpy as np
import pytest

import oracle
from oracle import mmio
from cask_amd import capi, synth
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_blas1_against_oracle():
    import torch
    rng = np.random.default_rng(1)
    for n in (1, 2, 63, 64, 1000, 1001, 262_144 + 7, 1_585_478):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        out = torch.zeros(1, dtype=torch.float64, device="cuda")
        capi.ddot_device(xt, yt, out)
        got = float(out.cpu()[0])
        want = oracle.ddot(x, y)
        assert abs(got - want) <= 1e-12 * max(1.0, np.abs(x * y).sum()), (n, got, want)
        out2 = torch.zeros(1, dtype=torch.float64, device="cuda")
        capi.ddot_device(xt, yt, out2)
        assert torch.equal(out, out2)                      # deterministic reduction

        num = torch.tensor([3.0], dtype=torch.float64, device="cuda")
        den = torch.tensor([-4.0], dtype=torch.float64, device="cuda")
        y1 = yt.clone()
        capi.daxpy_device(xt, y1, sign=-1.0, num_t=num, den_t=den)       # y += -1*(3/-4)*x
        np.testing.assert_allclose(y1.cpu().numpy(), oracle.daxpy(0.75, x, y), rtol=1e-15, atol=1e-15)
        y2 = yt.clone()
        capi.daxpby_device(2.0, xt, y2, beta=-0.5)
        np.testing.assert_allclose(y2.cpu().numpy(), oracle.daxpby(2.0, x, -0.5, y), rtol=1e-15, atol=1e-15)
        y3 = yt.clone()
        capi.daxpby_device(1.0, xt, y3, num_t=num, den_t=den)            # y = x + (3/-4) y
        np.testing.assert_allclose(y3.cpu().numpy(), oracle.daxpby(1.0, x, -0.75, y), rtol=1e-15, atol=1e-15)


def test_cg_known_answers(known_answers):
    for case in known_answers["cg_identity"]["cases"]:
        full = mmio.read_matrix(GOLDEN / case["matrix"])
        rhs = mmio.read_vector(GOLDEN / case["rhs"])
        m = capi.CsrMatrix.from_host(full.n, full.m, full.row_ptr, full.col_ind, full.values)
        x, iters, conv, _ = m.cg(rhs)
        m.close()
        assert conv
        want, want_it, _ = oracle.cg_full(full.row_ptr, full.col_ind, full.values, rhs)
        assert iters == want_it
        oracle.assert_almost_equal(x, np.array(case["expected"], dtype=float), rel=1e-12, abs_=1e-14,
                                   what=case["matrix"])


@pytest.mark.parametrize("name", ["cant", "G3_circuit"])
def test_cg_matches_oracle_on_spd_families(name):
    n, rp, ci, va = synth.small(name, factor=16)
    x0 = mmio.test_vector(n) / n
    b = oracle.csr_spmv(rp, ci, va, x0)                      # harness of test_utils.hpp:61-70
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    got, it, conv, us = m.cg(b)
    m.close()
    assert conv == want_conv
    assert abs(it - want_it) <= 2, (it, want_it)
    res = np.linalg.norm(b - oracle.csr_spmv(rp, ci, va, got))
    assert res <= 2e-5, res                                  # tol 1e-5 absolute on sqrt(r.r)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
    assert us > 0


def test_cg_not_converged_reports_last_iteration():
    n, rp, ci, va = synth.small("G3_circuit", factor=64)
    b = np.random.default_rng(7).standard_normal(n)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b, maxiters=5)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    got, it, conv, _ = m.cg(b, maxiters=5)
    m.close()
    assert not conv and not want_conv
    assert it == want_it == 4                                # iterations = i of the last pass (:231)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)


def test_bicg_matches_oracle_nonsymmetric():
    n, rp, ci, va = synth.small("atmosmodd", factor=16)
    x0 = mmio.test_vector(n) / n
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    got, it, conv, _ = m.bicg(b)
    m.close()
    assert conv and want_conv
    assert abs(it - want_it) <= 2, (it, want_it)
    res = np.linalg.norm(b - oracle.csr_spmv(rp, ci, va, got))
    assert res <= 2e-5
    np.testing.assert_allclose(got, x0, rtol=1e-5, atol=1e-6)


def test_bicg_reference_harness_bfwb62():
    """test/test_bicg.cpp:11 -> runTest(bfwb62): b = A x0, x0_i = 0.25 i, expect x0 back."""
    a = mmio.read_matrix(GOLDEN / "matrices/bfwb62.mtx")
    x0 = mmio.test_vector(a.n)
    b = oracle.csr_spmv(a.row_ptr, a.col_ind, a.values, x0)
    m = capi.CsrMatrix.from_host(a.n, a.m, a.row_ptr, a.col_ind, a.values)
    got, it, conv, _ = m.bicg(b, tol=1e-14, maxiters=500)
    m.close()
    assert conv
    np.testing.assert_allclose(got, x0, rtol=1e-6, atol=1e-8)


def test_config3_cg_on_full_g3_circuit_like():
    """BASELINE configs[2] at FULL size: CG on the 1.585 M-row G3_circuit-like SPD system, both
    pass forms: iterations within +-2 of the oracle's pcg restatement, true residual (oracle product) <= 2e-5."""
    import os
    n, rp, ci, va = synth.GENERATORS["G3_circuit"]()
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b)
    assert want_conv
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    xs = {}
    try:
        for mode in ("2", "1"):                                # classic (3 launches per pass), composed (2 launches)
            os.environ["CASK_HIP_SOLVER_MODE"] = mode
            got, it, conv, us = m.cg(b)
            assert conv and abs(it - want_it) <= 2, (mode, it, want_it)
            assert np.linalg.norm(b - oracle.csr_spmv(rp, ci, va, got)) <= 2e-5
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
            xs[mode] = (got, it)
    finally:
        os.environ.pop("CASK_HIP_SOLVER_MODE", None)
        m.close()
    # at full size the grid of the converging launch is not co-resident; every workgroup must still
    # apply the last x += alpha p (the convergence flag carries the pass that set it).  A dropped update shows up
    # as rows that differ between the two pass forms, which are otherwise identical to the bit.
    assert xs["1"][1] == xs["2"][1] and np.array_equal(xs["1"][0], xs["2"][0])


def test_config5_bicg_on_full_atmosmodd_like():
    """BASELINE configs[4] at FULL size on one GPU: BiCG with A and A^T on the 1.27 M-row nonsymmetric stencil system."""
    import os
    n, rp, ci, va = synth.GENERATORS["atmosmodd"]()
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b)
    assert want_conv
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    xs = {}
    try:
        for mode in ("2", "1"):
            os.environ["CASK_HIP_SOLVER_MODE"] = mode
            got, it, conv, us = m.bicg(b)
            assert conv and abs(it - want_it) <= 2, (mode, it, want_it)
            assert np.linalg.norm(b - oracle.csr_spmv(rp, ci, va, got)) <= 2e-5
            np.testing.assert_allclose(got, x0, rtol=1e-5, atol=1e-6)
            xs[mode] = (got, it)
    finally:
        os.environ.pop("CASK_HIP_SOLVER_MODE", None)
        m.close()
    assert xs["1"][1] == xs["2"][1] and np.array_equal(xs["1"][0], xs["2"][0])     # see the CG test above


def test_composed_and_classic_passes_agree_bit_for_bit():
    """The two-launch pass composes p = r + beta*p with the same fma the classic p update uses, sums the same
    partials in the same order and applies the same x update one launch later: identical iterates, not just close."""
    import os
    for name, solver in (("cant", "cg"), ("G3_circuit", "cg"), ("webbase-1M", "bicg"), ("atmosmodd", "bicg")):
        n, rp, ci, va = synth.small(name, factor=32)
        if solver == "bicg" and name == "webbase-1M":           # make it diagonally dominant so that BiCG converges
            rows = np.repeat(np.arange(n), np.diff(rp))
            va = va.copy()
            absum = np.zeros(n)
            np.add.at(absum, rows, np.abs(va))
            diag = rows == ci
            if not diag.any():
                continue
            va[diag] = absum[rows[diag]] + 1.0
        b = np.random.default_rng(9).standard_normal(n)
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
        out = {}
        try:
            for mode in ("1", "2"):
                os.environ["CASK_HIP_SOLVER_MODE"] = mode
                out[mode] = (m.cg if solver == "cg" else m.bicg)(b, maxiters=40, tol=1e-12)[:3]
        finally:
            os.environ.pop("CASK_HIP_SOLVER_MODE", None)
            m.close()
        assert out["1"][1:] == out["2"][1:], (name, out["1"][1:], out["2"][1:])
        assert np.array_equal(out["1"][0], out["2"][0]), name


def test_solvers_with_a_nonzero_initial_guess():
    for name, solver in (("cant", "cg"), ("atmosmodd", "bicg")):
        n, rp, ci, va = synth.small(name, factor=32)
        rng = np.random.default_rng(23)
        b, x_init = rng.standard_normal(n), rng.uniform(-1, 1, n)
        want, want_it, want_conv = (oracle.cg_full if solver == "cg" else oracle.bicg)(rp, ci, va, b, x0=x_init, tol=1e-9)
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
        got, it, conv, _ = (m.cg if solver == "cg" else m.bicg)(b, x0=x_init, tol=1e-9)
        m.close()
        assert conv == want_conv and abs(it - want_it) <= 2, (name, it, want_it)
        np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(want).max()))


def test_solvers_on_scan_plans():
    """CG / BiCG on handles planned with the SCAN variant: such a plan has no fused dot epilogue (the solver falls back to
    its own dot kernels)."""
    for name, solver in (("G3_circuit", "cg"), ("atmosmodd", "bicg")):
        n, rp, ci, va = synth.small(name, factor=16)
        b = np.random.default_rng(3).standard_normal(n)
        want, want_it, want_conv = (oracle.cg_full if solver == "cg" else oracle.bicg)(rp, ci, va, b, tol=1e-8)
        for dp in (dict(variant="scan", tile_width=1024), dict(variant="scan", tile_width=-1)):
            m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
            got, it, conv, _ = (m.cg if solver == "cg" else m.bicg)(b, tol=1e-8)
            m.close()
            assert conv == want_conv and abs(it - want_it) <= 2, (name, dp, it, want_it)
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(want).max()))



def _tridiagonal(n, lower, diag, upper):
    """A well-conditioned tridiagonal system in CSR (a handful of passes to convergence whatever n is)."""
    rows = np.repeat(np.arange(n), 3)[1:-1]
    cols = (rows + np.tile([-1, 0, 1], n)[1:-1])
    vals = np.tile([lower, diag, upper], n)[1:-1].astype(float)
    vals = vals * (1.0 + 0.01 * np.cos(np.arange(vals.size)))            # not a constant stencil
    rp = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp).astype(np.int32), cols.astype(np.int32), vals


@pytest.mark.parametrize("n", [524_288, 524_290, 524_291, 1_048_577, 2_097_154, 2_200_003])
def test_solver_launch_shapes_around_their_edges(n):
    """The update launches come in two shapes (blas1_kernels.hpp, r6): up to 1 024 workgroups of 256 threads with a pair
    per lane up to 262 144 pairs, 256 workgroups of 1 024 beyond -- there a lane requests its first 4 (3) pairs ahead of
    the scalars and walks the rest in a loop.  Sizes on both sides of the switch, odd and even, and past 4 pairs per lane
    (the BASELINE systems stay under 3.1), CG and BiCG, against the oracle; twice: the same bits."""
    for solver, (lo, up) in (("cg", (-1.0, -1.0)), ("bicg", (-1.3, -0.6))):
        rp, ci, va = _tridiagonal(n, lo, 4.0, up)
        if solver == "cg":                                               # symmetric: the upper entry of row i = the lower of row i + 1
            up_idx = np.arange(1, va.size - 1, 3)[: n - 1]
            va[up_idx] = va[up_idx + 1]
        x0 = np.cos(0.001 * np.arange(n))
        b = oracle.csr_spmv(rp, ci, va, x0)
        want, want_it, want_conv = (oracle.cg_full if solver == "cg" else oracle.bicg)(rp, ci, va, b, tol=1e-8)
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
        got, it, conv, _ = (m.cg if solver == "cg" else m.bicg)(b, tol=1e-8)
        again, it2, _, _ = (m.cg if solver == "cg" else m.bicg)(b, tol=1e-8)
        m.close()
        assert want_conv and conv and abs(it - want_it) <= 1, (solver, n, it, want_it)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(got, x0, rtol=1e-6, atol=1e-7)
        assert it2 == it and np.array_equal(got, again)
        if solver == "cg" and n in (524_291, 2_200_003):                 # ... and the Jacobi instantiations of the same launches
            want, want_it, want_conv = oracle.pcg_precond(rp, ci, va, b, kind="jacobi", tol=1e-8, full=True)
            pc = capi.Preconditioner("jacobi", n, rp, ci, va)
            m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
            got, it, conv, _ = m.pcg(pc, b, tol=1e-8)
            m.close()
            pc.close()
            assert want_conv and conv and abs(it - want_it) <= 1, ("pcg jacobi", n, it, want_it)
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
