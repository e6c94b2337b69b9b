"""Pin the CPU oracle: the reference's own known answers, the committed golden
vectors, scipy, and MKL (when the shared library is present).  CPU only."""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import mmio
from conftest import GOLDEN, golden_matrix_files


def dense_to_csr(rows):
    a = sp.csr_matrix(np.array(rows, dtype=np.float64))
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data


def ulp_close(a, b, ulps=4):
    """gtest ASSERT_DOUBLE_EQ: within 4 units in the last place."""
    a, b = np.float64(a), np.float64(b)
    return abs(a - b) <= ulps * np.spacing(max(abs(a), abs(b)))


# ---- known answers from test/SparseMatrix.cpp -------------------------------------

def test_csr_dot_known_answers(known_answers):
    for key in ("dok_dot", "csr_dot"):
        ka = known_answers[key]
        rp, ci, v = dense_to_csr(ka["dense_rows"])
        assert oracle.csr_spmv(rp, ci, v, ka["b"]).tolist() == ka["expected"]


def test_symcsr_dot_known_answer(known_answers):
    ka = known_answers["symcsr_dot"]
    rp, ci, v = dense_to_csr(ka["lower_dense_rows"])
    assert oracle.symcsr_spmv(rp, ci, v, ka["b"]).tolist() == ka["expected"]


def test_transpose_product_matches_scipy():
    rng = np.random.default_rng(5)
    a = sp.random(37, 23, 0.2, format="csr", random_state=rng)
    a.sort_indices()
    x = rng.standard_normal(37)
    y = oracle.csr_spmv_t(23, a.indptr, a.indices, a.data, x)
    np.testing.assert_allclose(y, a.T @ x, rtol=1e-13, atol=1e-14)


# ---- MatrixMarket restatement vs test/Io.cpp ---------------------------------------

def test_io_dense_4_exact_values(known_answers):
    ka = known_answers["io_dense_4"]
    m = mmio.read_matrix(GOLDEN / ka["file"])
    assert (m.n, m.m, m.nnz) == (ka["n"], ka["m"], ka["nnz"])
    dense = sp.csr_matrix((m.values, m.col_ind, m.row_ptr), shape=(m.n, m.m)).toarray()
    assert dense.tolist() == ka["dense_rows"]          # exact, as EXPECT_EQ in Io.cpp


def test_io_header(known_answers):
    ka = known_answers["io_header"]
    h = mmio.read_header(GOLDEN / ka["file"])
    assert (h.type, h.format, h.data_type, h.symmetry) == (
        ka["type"], ka["format"], ka["data_type"], ka["symmetry"])


def test_io_sym_expansion(known_answers):
    for case in known_answers["io_sym"]["cases"]:
        low = mmio.read_sym_matrix(GOLDEN / case["file"])
        assert (low.n, low.m) == (case["n"], case["m"])
        full = mmio.read_matrix(GOLDEN / case["file"])
        assert full.nnz == case["sym_nnz"]
        dense = sp.csr_matrix((full.values, full.col_ind, full.row_ptr), shape=(4, 4)).toarray()
        assert dense.tolist() == case["expanded_dense_rows"]


def test_io_rejects_bad_header(tmp_path):
    p = tmp_path / "bad.mtx"
    p.write_text("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n")
    with pytest.raises(ValueError, match="Not a valid MatrixMarket"):
        mmio.read_header(p)
    with pytest.raises(ValueError, match="File not found"):
        mmio.read_header(tmp_path / "missing.mtx")
    g = tmp_path / "gen.mtx"
    g.write_text("%%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 2.0\n")
    with pytest.raises(ValueError, match="not symmetric"):
        mmio.read_sym_matrix(g)


def test_io_duplicate_entry_last_wins(tmp_path):
    p = tmp_path / "dup.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n2 2 5.0\n1 1 7.0\n")
    m = mmio.read_matrix(p)
    assert m.values.tolist() == [7.0, 5.0] and m.nnz == 2


def test_io_asymmetric_symmetric_file_rejected(tmp_path):
    p = tmp_path / "asym.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n2 2 2\n2 1 1.0\n1 2 3.0\n")
    with pytest.raises(ValueError, match="not symmetric"):
        mmio.read_matrix(p)


# ---- golden vectors ---------------------------------------------------------------

@pytest.mark.parametrize("key,path", golden_matrix_files(), ids=lambda v: v if isinstance(v, str) else "")
def test_oracle_reproduces_golden(key, path, expected_y):
    m = mmio.read_matrix(path)
    x = mmio.test_vector(m.m)
    y = oracle.csr_spmv(m.row_ptr, m.col_ind, m.values, x)
    assert np.array_equal(y, expected_y[key])          # bit-exact: same code, same order
    a = sp.csr_matrix((m.values, m.col_ind, m.row_ptr), shape=(m.n, m.m))
    oracle.assert_almost_equal(a @ x, y, what=f"scipy vs oracle on {key}")


def _mkl():
    for cand in (os.environ.get("MKLROOT", "") + "/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.1",
                 "/opt/conda/lib/libmkl_rt.so.2"):
        if os.path.exists(cand):
            os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
            return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    return None


@pytest.mark.parametrize("key", ["matrices/OPF_6000", "matrices/test_cage6", "benchmark/t2d_q9_A_01",
                                 "matrices/test_wa"])
def test_oracle_vs_mkl(key, expected_y):
    L = _mkl()
    if L is None:
        pytest.skip("MKL shared library not present")
    path = dict(golden_matrix_files())[key]
    m = mmio.read_matrix(path)
    x, y = mmio.test_vector(m.m), np.zeros(m.n)
    tr, nn = ctypes.c_char(b"N"), ctypes.c_int(m.n)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.mkl_cspblas_dcsrgemv(ctypes.byref(tr), ctypes.byref(nn), p(m.values), p(m.row_ptr),
                           p(m.col_ind), p(x), p(y))
    oracle.assert_almost_equal(y, expected_y[key], what=f"MKL vs golden on {key}")


# ---- tolerance function -----------------------------------------------------------

def test_almost_equal_semantics():
    assert oracle.almost_equal(1.0, 1.0)
    assert oracle.almost_equal(0.0, 5e-12)                  # inside abs 1e-11
    assert not oracle.almost_equal(0.0, 5e-11)
    assert oracle.almost_equal(1e6, 1e6 * (1 + 5e-9))       # inside rel 1e-8
    assert not oracle.almost_equal(1e6, 1e6 * (1 + 5e-8))
    assert not oracle.almost_equal(float("nan"), 1.0)
    cnt, first = oracle.mismatches([1, 2, 3], [1, 2.1, 3.5])
    assert (cnt, first) == (2, 1)


# ---- CG known answers (test/LinearSolvers.cpp) -------------------------------------

def test_pcg_identity_known_answers(known_answers):
    for case in known_answers["cg_identity"]["cases"]:
        low = mmio.read_sym_matrix(GOLDEN / case["matrix"])
        rhs = mmio.read_vector(GOLDEN / case["rhs"])
        x, iters, conv = oracle.pcg_identity(low.row_ptr, low.col_ind, low.values, rhs)
        assert conv
        for got, exp in zip(x, case["expected"]):
            assert ulp_close(got, exp), (case["matrix"], x)
        # the same system through the expanded matrix
        full = mmio.read_matrix(GOLDEN / case["matrix"])
        x2, _, conv2 = oracle.cg_full(full.row_ptr, full.col_ind, full.values, rhs)
        assert conv2 and all(ulp_close(g, e) for g, e in zip(x2, case["expected"]))


def test_pcg_iteration_reporting():
    """`iterations` is only written at the end of a non-converged pass
    (SparseLinearSolvers.hpp:231): identity system converges in pass 0 -> 0."""
    low = mmio.read_sym_matrix(GOLDEN / "systems/tiny.mtx")
    _, iters, conv = oracle.pcg_identity(low.row_ptr, low.col_ind, low.values, [1, 2, 3, 4])
    assert conv and iters == 0


def test_solver_harness_protocol_bfwb62():
    """test_utils.hpp:61-70,120-163: b = A x0 with x0_i = 0.25 i, solution must
    come back almost_equal to x0 (CG: bfwb62 is symmetric positive definite?  it
    is symmetric; use BiCG which the reference's test_bicg.cpp:11 targets)."""
    m = mmio.read_matrix(GOLDEN / "matrices/bfwb62.mtx")
    x0 = mmio.test_vector(m.n)
    b = oracle.csr_spmv(m.row_ptr, m.col_ind, m.values, x0)
    x, iters, conv = oracle.bicg(m.row_ptr, m.col_ind, m.values, b, tol=1e-14, maxiters=500)
    assert conv
    np.testing.assert_allclose(x, x0, rtol=1e-6, atol=1e-8)


def test_bicg_nonsymmetric_small():
    rng = np.random.default_rng(3)
    n = 60
    a = sp.random(n, n, 0.1, format="csr", random_state=rng) + sp.diags(np.full(n, 4.0))
    a = sp.csr_matrix(a)
    a.sort_indices()
    x0 = rng.standard_normal(n)
    b = a @ x0
    x, _, conv = oracle.bicg(a.indptr, a.indices, a.data, b, tol=1e-12)
    assert conv
    np.testing.assert_allclose(x, x0, rtol=1e-8, atol=1e-10)


# ---- DSE sweep order (test/TestUtils.cpp) ------------------------------------------

def test_sweep_order_known_answers(known_answers):
    ka = known_answers["sweep_order"]
    pts = oracle.sweep_order([tuple(r) for r in ka["ranges"]])
    as_pairs = [[p["numPipes"], p["frequency"]] for p in pts]
    assert as_pairs[:4] == ka["first_four"]
    assert as_pairs[15] == ka["after_15_from_start"]
    assert as_pairs[16:18] == ka["then"]
    assert len(pts) == 18                              # every point, last included


# ---- DFE stream format restatement (oracle/dfe_format.py vs the C decoder) -----------------

@pytest.mark.parametrize("arch", [(3, 32, 5), (2, 1024, 16), (6, 256, 3)])
def test_dfe_format_restatement_decodes_to_the_product(arch, expected_y):
    from oracle import dfe_format
    pipes, cache, width = arch
    files = dict(golden_matrix_files())
    for key in ("matrices/test_cage6", "matrices/test_tiny", "matrices/test_long_row", "matrices/test_wa",
                "matrices/test_non_multiple", "matrices/test_large_empty", "matrices/bfwb62"):
        m = mmio.read_matrix(files[key])
        x = mmio.test_vector(m.m)
        parts = dfe_format.preprocess(m.n, m.m, m.row_ptr, m.col_ind, m.values, pipes, cache, width)
        assert len(parts) == pipes
        xp = np.concatenate([x, np.zeros((-x.size) % cache)])
        ys = [oracle.partition_decode_spmv(p["n"], p["n_blocks"], cache, width, False, p["colptr"],
                                           np.frombuffer(p["records"].tobytes(), dtype=np.uint8), xp) for p in parts]
        y = ys[0] if m.n < pipes else np.concatenate(ys)
        oracle.assert_almost_equal(y, expected_y[key], what=f"{key} {arch}")
        for p in parts:                                   # Spmv.cpp:76-77: records padded per block to input_width
            assert p["records"].size % width == 0 and p["colptr"].size == p["n"] * p["n_blocks"]


# ---- preconditioning known answers (test/LinearSolvers.cpp:54-146, test/MklLayer.cpp:10-50) ------

def _dense_to_csr(rows):
    a = sp.csr_matrix(np.array(rows, dtype=np.float64))
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)


def test_ilu_factor_and_apply_known_answers(known_answers):
    c = known_answers["ilu"]["factor_cases"][0]
    rp, ci, va = _dense_to_csr(c["dense_rows"])
    f = oracle.ilu0(rp, ci, va)
    got = sp.csr_matrix((f, ci, rp), shape=(4, 4)).toarray()
    assert np.array_equal(got, np.array(c["factored_rows"], dtype=np.float64))          # ASSERT_EQ
    z = oracle.ilu_apply(rp, ci, f, c["apply_to"])
    assert all(ulp_close(g, e) for g, e in zip(z, c["apply_expected"])), z
    # ILUCompute: the explicitly symmetric tinysym factors to all ones in its own pattern
    c2 = known_answers["ilu"]["factor_cases"][1]
    m = mmio.read_matrix(GOLDEN / c2["matrix_explicit_symmetric"])
    assert list(m.row_ptr) == c2["factored_csr"]["row_ptr"] and list(m.col_ind) == c2["factored_csr"]["col_ind"]
    assert list(oracle.ilu0(m.row_ptr, m.col_ind, m.values)) == c2["factored_csr"]["values"]


ZERO_PIVOT = (np.array([0, 2, 5, 7], dtype=np.int32), np.array([0, 1, 0, 1, 2, 1, 2], dtype=np.int32),
              np.array([0.0, 2.0, 2.0, 4.0, 1.0, 1.0, 3.0]))        # [[0*, 2, .], [2, 4, 1], [., 1, 3]], (0,0) STORED as zero
ZERO_PIVOT_FACTORED = [0.0, 2.0, 2.0, 4.0, 1.0, 0.25, 2.75]


def test_ilu_skips_a_stored_zero_pivot_like_isnnz():
    """DokMatrix::isNnz (SparseMatrix.hpp:219-225) tests the VALUE: a stored zero on the diagonal is "not there", so
    ILUPreconditioner (SparseLinearSolvers.hpp:100-101) skips that pivot instead of dividing by it.  Hand-computed: row 1
    keeps its (1,0) entry, row 2 gets 1/4 and 3 - 1*(1/4)."""
    rp, ci, va = ZERO_PIVOT
    assert list(oracle.ilu0(rp, ci, va)) == ZERO_PIVOT_FACTORED


def test_unittrsolve_known_answers(known_answers):
    for c in known_answers["unittrsolve"]["cases"]:
        rp, ci, va = _dense_to_csr(c["dense_rows"])
        x = oracle.trsolve(rp, ci, va, c["rhs"], lower=c["lower"])
        assert list(x) == c["expected"], (c, x)                                             # ASSERT_EQ


def test_pcg_with_ilu_known_answer(known_answers):
    """CGSymWithILUPC: the preconditioner is built from the stored LOWER triangle, the iteration
    stagnates for all 2000 passes and the reference pins where it ends up (4 ULP)."""
    c = known_answers["ilu"]["pcg_ilu"]
    low = mmio.read_sym_matrix(GOLDEN / c["matrix"])
    rhs = mmio.read_vector(GOLDEN / c["rhs"])
    x, iters, conv = oracle.pcg_precond(low.row_ptr, low.col_ind, low.values, rhs, kind="ilu0")
    assert conv == c["converges"] and iters == 1999
    assert all(ulp_close(g, e) for g, e in zip(x, c["expected"])), [float(v).hex() for v in x]


def test_pcg_precond_solves_spd_systems():
    """On a diagonally dominant SPD matrix both preconditioners converge to the solution (the ILU is built
    from the stored LOWER triangle, as the reference's pcg does, so it is no better than Jacobi)."""
    n, rp, ci, va = synth_small_spd()
    low = sp.tril(sp.csr_matrix((va, ci, rp), shape=(n, n))).tocsr()
    low.sort_indices()
    x0 = np.random.default_rng(4).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    its = {}
    for kind in ("jacobi", "ilu0"):
        x, it, conv = oracle.pcg_precond(low.indptr, low.indices, low.data, b, kind=kind, tol=1e-10)
        assert conv
        np.testing.assert_allclose(x, x0, rtol=1e-7, atol=1e-9)
        its[kind] = it
    assert max(its.values()) < 200


def synth_small_spd(n=400, seed=6):
    rng = np.random.default_rng(seed)
    a = sp.random(n, n, 0.02, format="csr", random_state=rng)
    a = a + a.T + sp.diags(np.full(n, 1.0) + np.abs(a + a.T).sum(axis=1).A1)
    a = sp.csr_matrix(a)
    a.sort_indices()
    return n, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)


def test_matrix_dir_files_go_through_the_products_reader(tmp_path, monkeypatch):
    """$CASK_MATRIX_DIR/<name>.mtx (real SuiteSparse inputs for bench.py / the tools) is read by the product's own
    reader -- cask::io::readMatrixCached behind libCaskHip.so (cask_amd/hostio.py) -- not by scipy: one committed
    SuiteSparse fixture (OPF_6000, symmetric, stored lower triangle) stands in for a BASELINE file.  Same CSR as the
    oracle's reader, the golden product reproduced, and the binary cache written and reused."""
    import gzip
    import shutil
    from cask_amd import hostio, synth
    from oracle import mmio
    src = GOLDEN / "matrices" / "OPF_6000.mtx.gz"
    dst = tmp_path / "cant.mtx"
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        shutil.copyfileobj(f, g)
    monkeypatch.setenv("CASK_MATRIX_DIR", str(tmp_path))
    n, rp, ci, va, source = synth.load_or_make("cant")
    assert source == str(dst) and n == 29902 and ci.size == 302418          # 166 160 stored entries, mirrored
    want = mmio.read_matrix(dst)
    assert np.array_equal(rp, want.row_ptr) and np.array_equal(ci, want.col_ind) and np.array_equal(va, want.values)
    expected = np.load(GOLDEN / "spmv_expected.npz")
    key = [k for k in expected.files if "OPF_6000" in k][0]
    oracle.assert_almost_equal(oracle.csr_spmv(rp, ci, va, mmio.test_vector(n)), expected[key], what="OPF_6000 through IO.hpp")
    assert (tmp_path / "cant.mtx.csrbin").exists()
    again = hostio.read_matrix(dst, cached=True)                             # served by the cache
    assert np.array_equal(again[2], rp) and np.array_equal(again[4], va)
    with pytest.raises(ValueError):
        hostio.read_matrix(tmp_path / "missing.mtx")
