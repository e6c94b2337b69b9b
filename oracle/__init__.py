"""CPU oracle for the CASK SpMV hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``cask_amd/``, ``include/``)
never does: it fails loudly when its HIP library is missing instead of falling
back to anything in here.

The arithmetic lives in ``cask_oracle.c`` (plain C, compiled with gcc by
``oracle/Makefile``); this module is the ctypes/numpy face of it plus the
MatrixMarket restatement in ``oracle.mmio``.  Each function names the reference
lines it follows (paths relative to /root/reference).

Parity status: pinned against the known answers in the reference's own gtest
suites (test/SparseMatrix.cpp, test/Io.cpp, test/LinearSolvers.cpp incl. the ILU
factor / apply / pcg+ILU answers, test/MklLayer.cpp, test/TestUtils.cpp ->
tests/golden/known_answers.json) and cross-checked
against MKL, the CPU library the reference calls.  The reference host code
cannot be built in this image (no Eigen/Boost/dfe-snippets), so there is no
``oracle/_ref``.  BiCG is "parity unpinned": the reference never defines it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libcask_oracle.so"
if os.environ.get("CASK_ORACLE_LIB"):                  # `make asan`: the same restatement under ASan / UBSan (tests/test_host_cpp.py)
    _LIB_PATH = Path(os.environ["CASK_ORACLE_LIB"]).resolve()
_lib = None

REL_TOL = 1e-8   # test/test_utils.hpp:36
ABS_TOL = 1e-11  # test/test_utils.hpp:36


def build(force: bool = False) -> Path:
    """Compile cask_oracle.c with gcc (idempotent)."""
    src = _HERE / "cask_oracle.c"
    if os.environ.get("CASK_ORACLE_LIB"):
        return _LIB_PATH
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s",
                        "_build/libcask_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_LIB_PATH))
        i32, i64, dbl = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
        p = ctypes.c_void_p
        L.oracle_csr_spmv.argtypes = [i32, p, p, p, p, p]
        L.oracle_csr_spmv.restype = None
        L.oracle_csr_spmv_t.argtypes = [i32, i32, p, p, p, p, p]
        L.oracle_csr_spmv_t.restype = None
        L.oracle_symcsr_spmv.argtypes = [i32, p, p, p, p, p]
        L.oracle_symcsr_spmv.restype = ctypes.c_int
        L.oracle_almost_equal.argtypes = [dbl, dbl, dbl, dbl]
        L.oracle_almost_equal.restype = ctypes.c_int
        L.oracle_count_mismatches.argtypes = [i64, p, p, dbl, dbl, ctypes.POINTER(i64)]
        L.oracle_count_mismatches.restype = i64
        L.oracle_ddot.argtypes = [i64, p, p]
        L.oracle_ddot.restype = dbl
        L.oracle_daxpy.argtypes = [i64, dbl, p, p]
        L.oracle_daxpy.restype = None
        L.oracle_daxpby.argtypes = [i64, dbl, p, dbl, p]
        L.oracle_daxpby.restype = None
        for name in ("oracle_pcg_identity", "oracle_cg_full", "oracle_bicg"):
            f = getattr(L, name)
            f.argtypes = [i32, p, p, p, p, p, i32, dbl, ctypes.POINTER(i32)]
            f.restype = ctypes.c_int
        L.oracle_ilu0.argtypes = [i32, p, p, p]
        L.oracle_ilu0.restype = None
        L.oracle_trsolve.argtypes = [i32, p, p, p, ctypes.c_int, p, p]
        L.oracle_trsolve.restype = None
        L.oracle_ilu_apply.argtypes = [i32, p, p, p, p, p, p]
        L.oracle_ilu_apply.restype = None
        L.oracle_pcg_precond.argtypes = [i32, p, p, p, ctypes.c_int, ctypes.c_int, p, p, i32, dbl, ctypes.POINTER(i32)]
        L.oracle_pcg_precond.restype = ctypes.c_int
        L.oracle_partition_decode_spmv.argtypes = [i32, i32, i32, i32, i32, p, i64, p, i64, p, p]
        L.oracle_partition_decode_spmv.restype = ctypes.c_int
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def csr_spmv(row_ptr, col_ind, values, x) -> np.ndarray:
    """y = A x, sequential per row in stored order (SparseMatrix.hpp:255-264,422-424)."""
    row_ptr, col_ind, values, x = _i32(row_ptr), _i32(col_ind), _f64(values), _f64(x)
    n = row_ptr.size - 1
    y = np.empty(n, dtype=np.float64)
    lib().oracle_csr_spmv(n, _ptr(row_ptr), _ptr(col_ind), _ptr(values), _ptr(x), _ptr(y))
    return y


def csr_spmv_t(n_cols, row_ptr, col_ind, values, x) -> np.ndarray:
    """y = A^T x (no reference counterpart; see cask_oracle.c)."""
    row_ptr, col_ind, values, x = _i32(row_ptr), _i32(col_ind), _f64(values), _f64(x)
    n = row_ptr.size - 1
    y = np.empty(n_cols, dtype=np.float64)
    lib().oracle_csr_spmv_t(n, n_cols, _ptr(row_ptr), _ptr(col_ind), _ptr(values), _ptr(x), _ptr(y))
    return y


def symcsr_spmv(row_ptr, col_ind, values, x) -> np.ndarray:
    """y = A x with only the lower triangle stored (SparseMatrix.hpp:512-514)."""
    row_ptr, col_ind, values, x = _i32(row_ptr), _i32(col_ind), _f64(values), _f64(x)
    n = row_ptr.size - 1
    y = np.empty(n, dtype=np.float64)
    rc = lib().oracle_symcsr_spmv(n, _ptr(row_ptr), _ptr(col_ind), _ptr(values), _ptr(x), _ptr(y))
    if rc:
        raise MemoryError("oracle_symcsr_spmv")
    return y


def almost_equal(got: float, exp: float, rel=REL_TOL, abs_=ABS_TOL) -> bool:
    return bool(lib().oracle_almost_equal(float(got), float(exp), rel, abs_))


def mismatches(got, exp, rel=REL_TOL, abs_=ABS_TOL):
    """(count, first_bad_index) under cask::test::check (test/test_utils.hpp:29-41)."""
    got, exp = _f64(got), _f64(exp)
    if got.shape != exp.shape:
        raise ValueError(f"shape mismatch {got.shape} vs {exp.shape}")
    first = ctypes.c_int64(-1)
    cnt = lib().oracle_count_mismatches(got.size, _ptr(got), _ptr(exp), rel, abs_, ctypes.byref(first))
    return int(cnt), int(first.value)


def assert_almost_equal(got, exp, rel=REL_TOL, abs_=ABS_TOL, what=""):
    cnt, first = mismatches(got, exp, rel, abs_)
    if cnt:
        g, e = np.asarray(got).ravel()[first], np.asarray(exp).ravel()[first]
        raise AssertionError(f"{what}: {cnt} mismatches; first at {first}: got {g!r} exp {e!r}")


def ddot(a, b) -> float:
    a, b = _f64(a), _f64(b)
    return float(lib().oracle_ddot(a.size, _ptr(a), _ptr(b)))


def daxpy(alpha, x, y) -> np.ndarray:
    x, y = _f64(x), _f64(y).copy()
    lib().oracle_daxpy(x.size, float(alpha), _ptr(x), _ptr(y))
    return y


def daxpby(alpha, x, beta, y) -> np.ndarray:
    x, y = _f64(x), _f64(y).copy()
    lib().oracle_daxpby(x.size, float(alpha), _ptr(x), float(beta), _ptr(y))
    return y


def _solve(fn, row_ptr, col_ind, values, rhs, x0, maxiters, tol):
    row_ptr, col_ind, values, rhs = _i32(row_ptr), _i32(col_ind), _f64(values), _f64(rhs)
    n = row_ptr.size - 1
    x = np.zeros(n) if x0 is None else _f64(x0).copy()
    it = ctypes.c_int32(0)
    rc = fn(n, _ptr(row_ptr), _ptr(col_ind), _ptr(values), _ptr(rhs), _ptr(x),
            int(maxiters), float(tol), ctypes.byref(it))
    if rc < 0:
        raise MemoryError("oracle solver")
    return x, int(it.value), bool(rc)


def pcg_identity(row_ptr, col_ind, values, rhs, x0=None, maxiters=2000, tol=1e-5):
    """pcg<double, IdentityPreconditioner> on a LOWER-triangular symmetric CSR
    (SparseLinearSolvers.hpp:162-239).  Returns (x, iterations, converged)."""
    return _solve(lib().oracle_pcg_identity, row_ptr, col_ind, values, rhs, x0, maxiters, tol)


def cg_full(row_ptr, col_ind, values, rhs, x0=None, maxiters=2000, tol=1e-5):
    """Same recurrence on a full (symmetry-expanded) CSR."""
    return _solve(lib().oracle_cg_full, row_ptr, col_ind, values, rhs, x0, maxiters, tol)


def bicg(row_ptr, col_ind, values, rhs, x0=None, maxiters=2000, tol=1e-5):
    """Classical BiCG (parity unpinned: no reference body)."""
    return _solve(lib().oracle_bicg, row_ptr, col_ind, values, rhs, x0, maxiters, tol)


def ilu0(row_ptr, col_ind, values) -> np.ndarray:
    """Factored values in the input pattern: ILUPreconditioner::pc (SparseLinearSolvers.hpp:88-113)."""
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    a = _f64(values).copy()
    lib().oracle_ilu0(row_ptr.size - 1, _ptr(row_ptr), _ptr(col_ind), _ptr(a))
    return a


def trsolve(row_ptr, col_ind, values, rhs, lower=True) -> np.ndarray:
    """cask::mkl::unittrsolve (MklLayer.hpp:29-85): triangle + its diagonal, textbook substitution."""
    row_ptr, col_ind, values, rhs = _i32(row_ptr), _i32(col_ind), _f64(values), _f64(rhs)
    n = row_ptr.size - 1
    x = np.zeros(n, dtype=np.float64)
    lib().oracle_trsolve(n, _ptr(row_ptr), _ptr(col_ind), _ptr(values), int(bool(lower)), _ptr(rhs), _ptr(x))
    return x


def ilu_apply(row_ptr, col_ind, factored, r) -> np.ndarray:
    """ILUPreconditioner::apply (:143-151) given the factored values."""
    row_ptr, col_ind, factored, r = _i32(row_ptr), _i32(col_ind), _f64(factored), _f64(r)
    n = row_ptr.size - 1
    tmp, z = np.zeros(n), np.zeros(n)
    lib().oracle_ilu_apply(n, _ptr(row_ptr), _ptr(col_ind), _ptr(factored), _ptr(r), _ptr(tmp), _ptr(z))
    return z


def pcg_precond(row_ptr, col_ind, values, rhs, kind="ilu0", x0=None, maxiters=2000, tol=1e-5, full=False):
    """pcg<double, ILUPreconditioner> on a LOWER-triangular symmetric CSR (kind="jacobi": z = r/diag;
    kind="ilu0_unit": the ILU factors applied with a unit lower diagonal, the textbook way);
    full=True: the arrays hold the whole symmetric matrix (product and preconditioner see all of it).
    Returns (x, iterations, converged)."""
    row_ptr, col_ind, values, rhs = _i32(row_ptr), _i32(col_ind), _f64(values), _f64(rhs)
    n = row_ptr.size - 1
    x = np.zeros(n) if x0 is None else _f64(x0).copy()
    it = ctypes.c_int32(0)
    rc = lib().oracle_pcg_precond(n, _ptr(row_ptr), _ptr(col_ind), _ptr(values), {"ilu0": 0, "jacobi": 1, "ilu0_unit": 2}[kind], int(bool(full)), _ptr(rhs),
                                  _ptr(x), int(maxiters), float(tol), ctypes.byref(it))
    if rc < 0:
        raise MemoryError("oracle solver")
    return x, int(it.value), bool(rc)


def partition_decode_spmv(n_rows, n_blocks, cache_size, input_width, rle, colptr, records, x):
    """Evaluate one DFE-format partition (Spmv.cpp:42-107 layout) on the CPU."""
    colptr = _i32(colptr)
    records = np.ascontiguousarray(records, dtype=np.uint8)
    assert records.size % 12 == 0
    x = _f64(x)
    y = np.empty(n_rows, dtype=np.float64)
    rc = lib().oracle_partition_decode_spmv(n_rows, n_blocks, cache_size, input_width, int(bool(rle)),
                                            _ptr(colptr), colptr.size, _ptr(records),
                                            records.size // 12, _ptr(x), _ptr(y))
    if rc:
        raise ValueError(f"malformed partition stream (code {rc})")
    return y


def sweep_order(ranges):
    """Design points in the order ChainedParameterRange visits them
    (src/runtime/Utils.hpp:158-202; pinned by test/TestUtils.cpp:12-49): the
    FIRST parameter varies fastest.  ``ranges`` is a list of
    (name, start, stop, step); returns a list of dicts.  Unlike the reference's
    DSE loop (Dse.cpp:40-47, which never evaluates the final point) every
    point is returned."""
    names = [r[0] for r in ranges]
    vals = []
    for _, start, stop, step in ranges:
        v, cur = [], start
        while True:
            v.append(cur)
            if cur == stop:            # Parameter::hasNext is value != end (Utils.hpp:145-147)
                break
            if cur + step > stop:      # Parameter::next throws (Utils.hpp:136-137)
                raise ValueError("range does not land on its end value")
            cur += step
        vals.append(v)
    out, idx = [], [0] * len(ranges)
    while True:
        out.append({n: vals[k][idx[k]] for k, n in enumerate(names)})
        k = 0
        while k < len(ranges) and idx[k] == len(vals[k]) - 1:
            idx[k] = 0
            k += 1
        if k == len(ranges):
            return out
        idx[k] += 1
