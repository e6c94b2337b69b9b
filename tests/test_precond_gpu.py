"""Preconditioning on the device against the oracle and the reference's known answers:
ILU(0) factor values (test/LinearSolvers.cpp:79-123), ILUPreconditioner::apply (:125-146), unittrsolve
(test/MklLayer.cpp:10-50, exact), pcg<double, ILUPreconditioner> (:54-77) and the level-scheduled
triangular solves on matrices with thousands of dependency levels.

Tolerances: factor values are computed by the same statements in the same order on the host ->
bitwise equal to the oracle; triangular solves run the same per-row substitution but the device
contracts "s -= v*x" into an FMA -> the reference's almost_equal(1e-8, 1e-11) against the oracle
(exact on the known answers); pcg sums dots in a different (tree) order -> relative 1e-10 against the oracle at
convergence, and 1e-9 relative after the 2000 stagnating passes of the reference's ILU test."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import mmio
from cask_amd import capi, synth
from conftest import GOLDEN, REPO, have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a GPU")]


def dense_csr(rows):
    a = sp.csr_matrix(np.array(rows, dtype=np.float64))
    a.sort_indices()
    return a.shape[0], a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)


def test_ilu_known_answers(known_answers):
    c = known_answers["ilu"]["factor_cases"][0]
    n, rp, ci, va = dense_csr(c["dense_rows"])
    pc = capi.Preconditioner("ilu0", n, rp, ci, va)
    got = sp.csr_matrix((pc.factor_values(), ci, rp), shape=(n, n)).toarray()
    assert np.array_equal(got, np.array(c["factored_rows"], dtype=np.float64))
    assert list(pc.apply(c["apply_to"])) == c["apply_expected"]
    pc.close()
    c2 = known_answers["ilu"]["factor_cases"][1]
    m = mmio.read_matrix(GOLDEN / c2["matrix_explicit_symmetric"])
    pc = capi.Preconditioner("ilu0", m.n, m.row_ptr, m.col_ind, m.values)
    assert list(pc.factor_values()) == c2["factored_csr"]["values"]
    pc.close()


def test_unittrsolve_known_answers(known_answers):
    for c in known_answers["unittrsolve"]["cases"]:
        n, rp, ci, va = dense_csr(c["dense_rows"])
        assert list(capi.trsolve(n, rp, ci, va, c["rhs"], lower=c["lower"])) == c["expected"]


def test_pcg_ilu_known_answer(known_answers):
    c = known_answers["ilu"]["pcg_ilu"]
    low = mmio.read_sym_matrix(GOLDEN / c["matrix"])
    full = mmio.read_matrix(GOLDEN / c["matrix"])
    rhs = mmio.read_vector(GOLDEN / c["rhs"])
    pc = capi.Preconditioner("ilu0", low.n, low.row_ptr, low.col_ind, low.values)     # the stored triangle (:171)
    m = capi.CsrMatrix.from_host(full.n, full.m, full.row_ptr, full.col_ind, full.values)
    x, it, conv, _ = m.pcg(pc, rhs)
    assert not conv and it == 1999
    np.testing.assert_allclose(x, c["expected"], rtol=1e-9)
    m.close()
    pc.close()


def test_ilu_skips_a_stored_zero_pivot_like_isnnz():
    """The product's factorisation follows DokMatrix::isNnz (value != 0, SparseMatrix.hpp:219-225) like the oracle's:
    hand-computed answer of tests/test_oracle.py."""
    from test_oracle import ZERO_PIVOT, ZERO_PIVOT_FACTORED
    rp, ci, va = ZERO_PIVOT
    pc = capi.Preconditioner("ilu0", 3, rp, ci, va)
    assert list(pc.factor_values()) == ZERO_PIVOT_FACTORED
    pc.close()


@pytest.mark.parametrize("name", ["G3_circuit", "cant", "atmosmodd"])
def test_ilu_factor_and_solves_match_the_oracle(name):
    n, rp, ci, va = synth.small(name)
    pc = capi.Preconditioner("ilu0", n, rp, ci, va)
    f = pc.factor_values()
    assert np.array_equal(f, oracle.ilu0(rp, ci, va))
    info = pc.info()
    assert info["levels_lower"] > 1 and info["levels_upper"] > 1 and info["launches_per_apply"] >= 2
    r = np.random.default_rng(2).uniform(-1, 1, n)
    z = pc.apply(r)
    oracle.assert_almost_equal(z, oracle.ilu_apply(rp, ci, f, r), what="ILU apply")
    # the two triangles on their own
    for lower in (True, False):
        oracle.assert_almost_equal(capi.trsolve(n, rp, ci, f, r, lower=lower), oracle.trsolve(rp, ci, f, r, lower=lower),
                                   what=f"trsolve lower={lower}")
    pc.close()


def test_wide_levels_get_their_own_launch():
    """A diagonal block structure: one level with 40 000 independent rows, then a dependent level."""
    n = 60_000
    rows = np.concatenate([np.arange(n), np.arange(40_000, n)])
    cols = np.concatenate([np.arange(n), np.arange(0, 20_000)])
    vals = np.concatenate([np.full(n, 2.0), np.full(20_000, -1.0)])
    a = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    a.sort_indices()
    rp, ci, va = a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data
    b = np.random.default_rng(1).standard_normal(n)
    oracle.assert_almost_equal(capi.trsolve(n, rp, ci, va, b, lower=True), oracle.trsolve(rp, ci, va, b, lower=True))


@pytest.mark.parametrize("kind", ["jacobi", "ilu0_unit"])
def test_pcg_converges_like_the_oracle(kind):
    """Whole symmetric matrix for product and preconditioner; ILU applied the textbook way (the reference's
    application divides by the diagonal twice: pinned by its known answer above and by the few-pass test
    below -- PCG stagnates with it)."""
    n, rp, ci, va = synth.small("G3_circuit")
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, it_want, conv_want = oracle.pcg_precond(rp, ci, va, b, kind=kind, tol=1e-9, full=True)
    pc = capi.Preconditioner(kind, n, rp, ci, va)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    x, it, conv, us = m.pcg(pc, b, tol=1e-9)
    assert conv and conv_want and abs(it - it_want) <= 2, (conv, conv_want, it, it_want)
    np.testing.assert_allclose(x, want, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(x, x0, rtol=1e-6, atol=1e-8)
    m.close()
    pc.close()


def test_reference_ilu_recurrence_first_passes():
    """The reference's ILU application on a real system: compare the iterate after 6 passes with the oracle."""
    n, rp, ci, va = synth.small("G3_circuit")
    a = sp.csr_matrix((va, ci, rp), shape=(n, n))
    low = sp.tril(a).tocsr()
    low.sort_indices()
    b = oracle.csr_spmv(rp, ci, va, np.random.default_rng(8).uniform(-1, 1, n))
    want, _, _ = oracle.pcg_precond(low.indptr, low.indices, low.data, b, kind="ilu0", maxiters=6, tol=1e-30)
    pc = capi.Preconditioner("ilu0", n, low.indptr, low.indices, low.data)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    x, it, conv, _ = m.pcg(pc, b, maxiters=6, tol=1e-30)
    assert not conv and it == 5
    np.testing.assert_allclose(x, want, rtol=1e-9, atol=1e-12)
    m.close()
    pc.close()


def test_pcg_with_textbook_ilu_beats_plain_cg():
    """ILU(0) of the whole matrix applied with a unit lower diagonal is a real preconditioner."""
    n, rp, ci, va = synth.small("G3_circuit")
    x0 = np.random.default_rng(6).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    _, it_plain, conv_plain, _ = m.cg(b, tol=1e-9)
    pc = capi.Preconditioner("ilu0_unit", n, rp, ci, va)
    x, it_ilu, conv_ilu, _ = m.pcg(pc, b, tol=1e-9)
    assert conv_plain and conv_ilu and it_ilu < it_plain
    np.testing.assert_allclose(x, x0, rtol=1e-6, atol=1e-8)
    m.close()
    pc.close()


def test_pcg_rejects_a_preconditioner_of_another_order():
    n, rp, ci, va = synth.small("cant")
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    pc = capi.Preconditioner("jacobi", 2, [0, 1, 2], [0, 1], [1.0, 1.0])
    with pytest.raises(ValueError, match="different order"):
        m.pcg(pc, np.ones(n))
    pc.close()
    m.close()


def test_argument_checks():
    with pytest.raises(ValueError, match="ascending"):
        capi.Preconditioner("ilu0", 2, [0, 2, 2], [1, 0], [1.0, 2.0])
    with pytest.raises(ValueError, match="kind"):
        capi.Preconditioner(7, 1, [0, 1], [0], [1.0])


def _lower_system(n, deps, seed=0):
    """Lower-triangular CSR (diagonal 2..3) with the strictly-lower columns `deps(i)` in row i."""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for i in range(n):
        cs = sorted(set(int(c) for c in deps(i) if 0 <= c < i)) + [i]
        rows += [i] * len(cs)
        cols += cs
    vals = rng.uniform(-0.4, 0.4, len(cols))
    a = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    a.setdiag(rng.uniform(2.0, 3.0, n))
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.copy()


@pytest.mark.parametrize("case", ["one_row", "diagonal", "chain", "chain_far_back", "comb", "alternating_widths", "fan_in"])
def test_packed_walk_edge_cases_match_the_oracle(case):
    """Shapes that stress the packed walk of the triangular solves (cask_hip_precond.hip): a single row; no dependencies
    at all (one wide level); a pure chain of 5000 levels of one row (several chunks, every dependency in the LDS
    ring); the same chain with a second dependency 3000 / 4500 rows back (beyond what the ring may serve: loaded with
    the chunk, a chunk ahead); a comb of 64-row levels; levels that alternate between 1 and 300 rows (packed steps and
    per-level launches interleaved); rows that depend on 40 earlier rows each (long-row variant).  Lower solve, and the
    transposed system as an upper solve."""
    rng = np.random.default_rng(11)
    if case == "one_row":
        n, deps = 1, lambda i: []
    elif case == "diagonal":
        n, deps = 5000, lambda i: []
    elif case == "chain":
        n, deps = 5000, lambda i: [i - 1]
    elif case == "chain_far_back":
        n, deps = 9000, lambda i: [i - 1, i - 3000, i - 4500]
    elif case == "comb":
        n, deps = 64 * 120, lambda i: [i - 64, i - 128] if i >= 64 else []
    elif case == "alternating_widths":
        # blocks of 301 rows: a head row that depends on the previous block's head, then 300 rows that depend on it
        n, deps = 301 * 40, lambda i: [i - 301] if i % 301 == 0 else [i - i % 301]
    else:
        n, deps = 3000, lambda i: list(rng.integers(0, max(i, 1), 40))
    rp, ci, va = _lower_system(n, deps)
    b = rng.standard_normal(n)
    oracle.assert_almost_equal(capi.trsolve(n, rp, ci, va, b, lower=True), oracle.trsolve(rp, ci, va, b, lower=True),
                               what=f"{case} lower")
    at = sp.csr_matrix((va, ci, rp), shape=(n, n)).T.tocsr()
    at.sort_indices()
    urp, uci, uva = at.indptr.astype(np.int32), at.indices.astype(np.int32), at.data
    oracle.assert_almost_equal(capi.trsolve(n, urp, uci, uva, b, lower=False), oracle.trsolve(urp, uci, uva, b, lower=False),
                               what=f"{case} upper")


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_packed_walk_random_dependency_graphs(seed):
    """Random lower-triangular systems: every row depends on 0-6 earlier rows at mixed distances (next door, a few
    thousand rows back, anywhere), so that levels of every width occur and a row's dependencies come from the LDS ring,
    from the chunk-ahead loads and from earlier launches alike.  Against the oracle, lower and (transposed) upper."""
    rng = np.random.default_rng(seed)
    n = 30_000 + 1000 * seed

    def deps(i):
        k = int(rng.integers(0, 7))
        reach = rng.choice([3, 60, 2500, 6000, max(i, 1)], size=k)
        return [i - 1 - int(rng.integers(0, max(1, min(int(r), i)))) for r in reach] if i else []

    rp, ci, va = _lower_system(n, deps, seed=seed)
    b = rng.standard_normal(n)
    oracle.assert_almost_equal(capi.trsolve(n, rp, ci, va, b, lower=True), oracle.trsolve(rp, ci, va, b, lower=True),
                               what="random lower")
    at = sp.csr_matrix((va, ci, rp), shape=(n, n)).T.tocsr()
    at.sort_indices()
    urp, uci, uva = at.indptr.astype(np.int32), at.indices.astype(np.int32), at.data
    oracle.assert_almost_equal(capi.trsolve(n, urp, uci, uva, b, lower=False), oracle.trsolve(urp, uci, uva, b, lower=False),
                               what="random upper")


def test_triangular_solve_schedules_agree_bit_for_bit(tmp_path):
    """Schedules of the same triangular solve (cask_hip_precond.hip): walker + stagers in position space (walk2), the
    four-wave packed walk of narrow-level runs (packed, r2) and the row-indexed walk of round 1
    (CASK_HIP_TRSV=levels).  (The one-walker-wave kernel and the one-launch synchronisation-free solve, measured losses,
    left the engine in round 5.)  All walk every row in stored order: identical bits -- on a grid factor with thousands
    of levels and long-range edges, a 3-D stencil, a banded FEM-like factor with 40 entries per row (several chunks
    per level run, entry-capped chunks), an arrow matrix whose last row is longer than a chunk can hold (that
    step falls back to the row-indexed walk), a ragged random triangle (rows of 0-9 entries, sources near, beyond
    the LDS ring and anywhere) and a dense band (rows of 1 to ~250 entries).  The lane-group walk (r5: a row on 1-64
    lanes, 4 / 8 / 16 entries a lane) adds in another order: the same solution to rounding."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, ".")
from cask_amd import capi, synth
out, levels = [], []
rng = np.random.default_rng(3)
def arrow(n):
    rows = [[i] for i in range(n - 1)] + [list(range(n))]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
    ci = np.concatenate(rows).astype(np.int32)
    va = rng.uniform(0.5, 1.5, ci.size)
    va[rp[1:] - 1] += 4.0
    return n, rp, ci, va
def ragged(n):
    """rows of 0..9 lower entries (Poisson 2.6): records with absent entries, rows longer than a record holds, sources
    a few rows back (deep levels), a few thousand back (beyond the LDS ring: read from memory) and anywhere"""
    rows = []
    for i in range(n):
        k = min(int(rng.poisson(2.6)), i)
        near = i - 1 - rng.geometric(0.3, size=k)
        far = rng.integers(0, max(i, 1), size=k)
        mid = i - rng.integers(4000, 9000, size=k)
        pick = rng.random(k)
        cols = np.where(pick < 0.6, near, np.where(pick < 0.8, mid, far))
        cols = np.unique(cols[(cols >= 0) & (cols < i)])
        rows.append(np.concatenate([cols, [i]]))
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
    ci = np.concatenate(rows).astype(np.int32)
    va = rng.uniform(-1.0, 1.0, ci.size)
    va[rp[1:] - 1] = rng.uniform(2.0, 3.0, n) * rng.choice([-1.0, 1.0], n)
    return n, rp, ci, va
cases = [synth.small("G3_circuit", factor=16), synth.small("atmosmodd", factor=32),
         synth.cant_like(n=6000, per_row=41, band=300, seed=2), arrow(5000), ragged(30000),
         synth.cant_like(n=800, per_row=350, band=400, seed=5)]        # rows of 1 .. ~250 entries: lane groups of 1 .. 64
rhs = [rng.standard_normal(c[0]) for c in cases]
# the schedule is a property of a factor, read from CASK_HIP_TRSV when it is built: one process, every schedule
for mode in sys.argv[2:]:
    os.environ.pop("CASK_HIP_TRSV", None)
    os.environ.pop("CASK_HIP_TRSV_LANES_E", None)
    if mode.startswith("lanes") and mode != "lanes":          # lanes4 / lanes16: that many entries per lane, whatever is cheapest
        os.environ["CASK_HIP_TRSV"], os.environ["CASK_HIP_TRSV_LANES_E"] = "lanes", mode[5:]
    elif mode != "default":
        os.environ["CASK_HIP_TRSV"] = mode
    print("MODE", mode, file=sys.stderr, flush=True)
    out, levels = [], []
    for case_no, (n, rp, ci, va) in enumerate(cases):
        r = rhs[case_no]
        if case_no == 4:                                       # a triangular matrix as it stands: no factorisation
            out.append(capi.trsolve(n, rp, ci, va, r, lower=True))
            continue
        for kind in ("ilu0_unit", "ilu0"):
            pc = capi.Preconditioner(kind, n, rp, ci, va)
            out.append(pc.apply(r))
            levels.append(pc.info()["levels_lower"])
            pc.close()
        out.append(capi.trsolve(n, rp, ci, va, r, lower=True))
        out.append(capi.trsolve(n, rp, ci, va, r, lower=False))
    np.save(sys.argv[1] + mode + ".npy", np.concatenate(out))
    print(max(levels))
'''
    # packed = the four-wave walk (r2); walk2 = r3; lanes = the lane-group walk for every run of levels that qualifies (r5);
    # default = lanes where the rows are long (the FEM-like factor), walk2 elsewhere
    modes = ("levels", "packed", "walk2", "lanes", "lanes4", "lanes16", "default")
    stem = str(tmp_path / "cask_trsv_")                              # (r6, ADVICE r5: fixed /tmp names collided across concurrent suites)
    res = subprocess.run([sys.executable, "-c", code, stem, *modes], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, CASK_HIP_TRSV_STATS="1"), cwd=str(REPO))
    assert res.returncode == 0, res.stderr[-1500:]
    assert all(int(line) > 500 for line in res.stdout.strip().splitlines()[-len(modes):])
    outs = {("" if m == "default" else m): np.load(f"{stem}{m}.npy") for m in modes}
    stderr = {}
    for part in res.stderr.split("MODE ")[1:]:
        name, _, text = part.partition("\n")
        stderr["" if name.strip() == "default" else name.strip()] = text
    assert np.all(np.isfinite(outs["levels"]))
    assert np.array_equal(outs["levels"], outs["packed"])
    assert np.array_equal(outs["levels"], outs["walk2"])
    # The lane-group walk adds a row's products group by group, not in stored order: the same solution to rounding.  It
    # must really have run: everywhere it can under `lanes` (wherever every source is inside the ring and no row exceeds 1 024 entries), on the FEM-like
    # factor alone by default.
    import re
    def lanes_chunks(text):
        return [int(m) for m in re.findall(r"lanes [LU]: \d+ runs of narrow levels, (\d+) chunks", text)]
    n_forced, n_default = (sum(c > 0 for c in lanes_chunks(stderr[m])) for m in ("lanes", ""))
    print("factors with lane-group runs: forced", n_forced, "default", n_default)
    assert n_forced >= 6 and 0 < n_default <= n_forced, (n_forced, n_default, stderr["lanes"][-800:])
    scale = np.abs(outs["levels"]).max()
    for mode in ("lanes", "lanes4", "lanes16", ""):
        assert np.all(np.isfinite(outs[mode]))
        assert np.abs(outs[mode] - outs["levels"]).max() <= 1e-11 * scale, (mode, np.abs(outs[mode] - outs["levels"]).max(), scale)
    assert not np.array_equal(outs["lanes"], outs["levels"])            # (a different order of additions)


def test_removed_multicolour_ilu_is_rejected_with_a_reason():
    """ABI 7: CASK_HIP_PRECOND_ILU0_MC (kind 4: ILU(0) of the colour-permuted matrix, not the reference's
    factors, behind Jacobi end to end -- docs/experiments.md) left the shipped engine; asking for it names the replacements."""
    with pytest.raises(ValueError, match="removed in ABI 7"):
        capi.Preconditioner(4, 2, [0, 2, 4], [0, 1, 0, 1], [2.0, 1.0, 1.0, 2.0])
