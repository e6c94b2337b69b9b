"""MatrixMarket ingest through the product's own reader (``cask::io::readMatrix`` / ``readMatrixCached``,
include/cask/IO.hpp -- the reference's io::readMatrix, src/runtime/IO.hpp:151-163: counting-sort COO -> CSR,
symmetric expansion, last duplicate wins, binary cache beside the text file), via three C entry points of
``libCaskHip.so``.  Plumbing for bench.py and the tools: real SuiteSparse files in ``$CASK_MATRIX_DIR`` take this
path, not scipy's."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libCaskHip.so"
_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build())")
        L = ctypes.CDLL(str(LIB_PATH))
        L.cask_host_last_error.restype = c_char_p
        L.cask_host_free.argtypes = [c_void_p]
        L.cask_host_read_matrix.argtypes = [c_char_p, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int64),
                                            POINTER(POINTER(c_int32)), POINTER(POINTER(c_int32)), POINTER(POINTER(c_double))]
        L.cask_host_read_matrix.restype = c_int
        _lib = L
    return _lib


def read_matrix(path, cached=True):
    """(n_rows, n_cols, row_ptr[int32], col_ind[int32], values[float64]) of a MatrixMarket coordinate file; raises
    ValueError with the reader's message (the reference's exception texts) on malformed input."""
    L = load()
    n, m, nnz = c_int32(0), c_int32(0), c_int64(0)
    rp, ci, va = POINTER(c_int32)(), POINTER(c_int32)(), POINTER(c_double)()
    if L.cask_host_read_matrix(str(path).encode(), 1 if cached else 0, byref(n), byref(m), byref(nnz), byref(rp), byref(ci),
                               byref(va)) != 0:
        raise ValueError(L.cask_host_last_error().decode(errors="replace"))
    try:
        row_ptr = np.ctypeslib.as_array(rp, shape=(n.value + 1,)).copy()
        col_ind = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),))[: nnz.value].copy()
        values = np.ctypeslib.as_array(va, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    finally:
        for p in (rp, ci, va):
            L.cask_host_free(ctypes.cast(p, c_void_p))
    return n.value, m.value, row_ptr, col_ind, values
