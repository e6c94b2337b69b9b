"""MatrixMarket reading as the reference does it -- TEST INFRASTRUCTURE ONLY.

Restates src/runtime/IO.hpp (paths relative to /root/reference):
  readHeader   :60-71   first line must match the regex on :66 exactly
  readDokMatrix:124-148 skip '%' lines, "n m l", then l whitespace-separated
                        (i j val) triples, 1-based, DokMatrix::set = last wins
                        (SparseMatrix.hpp:213-217)
  readMatrix   :151-163 symmetric files are mirrored explicitly
                        (DokMatrix::explicitSymmetric, SparseMatrix.hpp:156-189)
  readSymMatrix:165-176 keeps only what the file holds (the lower triangle)
  readVector   :73-116  "array" files: n values after the size line
CSR layout follows CsrMatrix(const DokMatrix&) (SparseMatrix.hpp:289-305):
rows in order, columns ascending inside a row, explicit zeros kept.

One deliberate difference: the reference counts nnzs once per file entry even
when an entry repeats (set() increments blindly, :216), which leaves
row_ptr.back() != col_ind.size() for files with duplicates; here nnz is the
number of stored entries.  No fixture has duplicates.
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass

import numpy as np

_HEADER_RE = re.compile(
    r"%%MatrixMarket (matrix|array) (coordinate|array) (real|integer) (symmetric|general)")


@dataclass
class MmInfo:
    type: str
    format: str
    data_type: str
    symmetry: str

    def is_symmetric(self):
        return self.symmetry == "symmetric"


@dataclass
class Csr:
    n: int
    m: int
    row_ptr: np.ndarray
    col_ind: np.ndarray
    values: np.ndarray

    @property
    def nnz(self):
        return int(self.col_ind.size)


def _open(path):
    path = str(path)
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "rt")


def read_header(path) -> MmInfo:
    try:
        with _open(path) as f:
            first = f.readline().rstrip("\n")
    except FileNotFoundError:
        raise ValueError(f"File not found {path}")
    m = _HEADER_RE.fullmatch(first)
    if not m:
        raise ValueError(f"Not a valid MatrixMarket file in {path}")
    return MmInfo(*m.groups())


def _read_coo(path):
    with _open(path) as f:
        line = f.readline()
        while line and line.startswith("%"):
            line = f.readline()
        n, m, l = (int(t) for t in line.split()[:3])
        toks = f.read().split()
    toks = toks[: 3 * l]
    if len(toks) < 3 * l:
        raise ValueError("File has less than given nonzeros!")
    i = np.array(toks[0::3], dtype=np.int64) - 1
    j = np.array(toks[1::3], dtype=np.int64) - 1
    v = np.array(toks[2::3], dtype=np.float64)
    return n, m, i, j, v


def _coo_to_csr_last_wins(n, m, i, j, v) -> Csr:
    if i.size:
        if i.min() < 0 or i.max() >= n or j.min() < 0 or j.max() >= m:
            raise ValueError("entry outside the declared shape")
    key = i * m + j
    order = np.argsort(key, kind="stable")
    key, v = key[order], v[order]
    # last occurrence of each key wins (DokMatrix::set overwrites)
    last = np.ones(key.size, dtype=bool)
    last[:-1] = key[1:] != key[:-1]
    key, v = key[last], v[last]
    rows = (key // m).astype(np.int64)
    cols = (key % m).astype(np.int32)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    return Csr(n, m, row_ptr, cols, v.astype(np.float64))


def read_matrix(path) -> Csr:
    """io::readMatrix: full CSR, symmetric files expanded."""
    info = read_header(path)
    if info.type != "matrix":
        raise ValueError(f"Error! Expecting MatrixMarket matrix in {path}")
    n, m, i, j, v = _read_coo(path)
    if info.is_symmetric():
        # explicitSymmetric: keep (i,j), add (j,i) for i != j; a file that holds
        # both (i,j) and (j,i) with different values is rejected (:172-175).
        base = _coo_to_csr_last_wins(n, m, i, j, v)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(base.row_ptr))
        cols = base.col_ind.astype(np.int64)
        off = rows != cols
        lookup = dict(zip(zip(rows.tolist(), cols.tolist()), base.values.tolist()))
        for r, c, val in zip(rows[off].tolist(), cols[off].tolist(), base.values[off].tolist()):
            other = lookup.get((c, r))
            if other is not None and other != val:
                raise ValueError("Matrix is not symmetric")
        i2 = np.concatenate([rows, cols[off]])
        j2 = np.concatenate([cols, rows[off]])
        v2 = np.concatenate([base.values, base.values[off]])
        return _coo_to_csr_last_wins(n, m, i2, j2, v2)
    return _coo_to_csr_last_wins(n, m, i, j, v)


def read_sym_matrix(path) -> Csr:
    """io::readSymMatrix: the stored (lower) triangle only; rejects general files."""
    info = read_header(path)
    if info.type != "matrix":
        raise ValueError(f"Error! Expecting MatrixMarket matrix in {path}")
    if not info.is_symmetric():
        raise ValueError(f"Error! Matrix found in {path} is not symmetric.")
    n, m, i, j, v = _read_coo(path)
    return _coo_to_csr_last_wins(n, m, i, j, v)


def read_vector(path) -> np.ndarray:
    """io::readVector for 'array' files (the only kind the reference's tests use)."""
    info = read_header(path)
    if info.format != "array":
        raise ValueError("only array-format vectors are restated")
    with _open(path) as f:
        line = f.readline()
        while line and line.startswith("%"):
            line = f.readline()
        n = int(line.split()[0])
        vals = f.read().split()[:n]
    return np.array(vals, dtype=np.float64)


def test_vector(n) -> np.ndarray:
    """x_i = 0.25 * i, the operand of test/test_spmv.cpp:27-28."""
    return np.arange(n, dtype=np.float64) * 0.25
