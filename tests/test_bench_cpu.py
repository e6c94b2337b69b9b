"""CPU-side pieces of bench.py: the MKL baselines the bench line reports next to the GPU number (the reference's own
CPU path: mkl_cspblas_dcsrgemv / mkl_dcsrsymv('l') + cblas, SparseLinearSolvers.hpp:162-239, fpgaNaiveCpuCode.cpp:33).
No GPU involved; skipped where the MKL runtime is not installed."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")

import bench  # noqa: E402
import oracle  # noqa: E402
from cask_amd import synth  # noqa: E402

mkl = bench.load_mkl()
needs_mkl = pytest.mark.skipif(mkl is None, reason="no MKL runtime in this environment")


def test_symmetry_check_and_stored_triangle():
    n, rp, ci, va = synth.small("G3_circuit", factor=64)
    assert bench.is_symmetric(rp, ci, va)
    lrp, lci, lva = bench.lower_triangle_1based(rp, ci, va)
    assert lrp[0] == 1 and lrp[-1] == lci.size + 1 and lva.size == lci.size
    rows = np.repeat(np.arange(n), np.diff(lrp))
    assert np.all(lci - 1 <= rows)                            # lower triangle + diagonal, 1-based
    assert lci.size == (ci.size - n) // 2 + n                 # every off-diagonal pair once, the diagonal once
    n2, rp2, ci2, va2 = synth.small("atmosmodd", factor=64)
    assert not bench.is_symmetric(rp2, ci2, va2)


@needs_mkl
def test_mkl_products_of_the_baseline_match_the_oracle():
    n, rp, ci, va = synth.small("G3_circuit", factor=64)
    x = np.random.default_rng(0).standard_normal(n)
    want = oracle.csr_spmv(rp, ci, va, x)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    nn, tr, lo = ctypes.c_int(n), ctypes.c_char(b"N"), ctypes.c_char(b"l")
    y = np.zeros(n)
    mkl.mkl_cspblas_dcsrgemv(ctypes.byref(tr), ctypes.byref(nn), p(va), p(rp), p(ci), p(x), p(y))
    oracle.assert_almost_equal(y, want, what="mkl_cspblas_dcsrgemv")
    lrp, lci, lva = bench.lower_triangle_1based(rp, ci, va)
    ys = np.zeros(n)
    mkl.mkl_dcsrsymv(ctypes.byref(lo), ctypes.byref(nn), p(lva), p(lrp), p(lci), p(x), p(ys))
    oracle.assert_almost_equal(ys, want, what="mkl_dcsrsymv('l')")


@needs_mkl
@pytest.mark.parametrize("kind,name", [("cg", "G3_circuit"), ("bicg", "atmosmodd")])
def test_solver_baseline_reports_both_routines(kind, name):
    n, rp, ci, va = synth.small(name, factor=64)
    b = oracle.csr_spmv(rp, ci, va, np.ones(n))
    out = bench.cpu_baseline_solver(kind, rp, ci, va, b, 1.0)
    assert out["kind"] == "mkl" and out["value"] > 0 and out["port"]["kind"] == "port"
    routines = set(out["gflops_by_routine_and_threads"])
    assert "mkl_cspblas_dcsrgemv" in routines
    assert ("mkl_dcsrsymv('l')" in routines) == (kind == "cg")


@needs_mkl
def test_spmv_baseline_times_the_symmetric_routine_for_symmetric_matrices():
    n, rp, ci, va = synth.small("cant", factor=16)
    x = np.arange(n) * 0.25 / n
    out = bench.cpu_baseline(rp, ci, va, x, None, 1.0)
    assert out["kind"] == "mkl" and out["mismatches_vs_oracle"] == 0
    assert set(out["gflops_by_routine_and_threads"]) == {"mkl_cspblas_dcsrgemv", "mkl_dcsrsymv('l')"}
    assert out["port"]["cores"] == 1
