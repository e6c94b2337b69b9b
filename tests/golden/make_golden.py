#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own test DATA (run in the build
container only; /root/reference does not exist on the GPU box).

What it does
  1. copies the MatrixMarket data files the reference's tests hold
     (test/matrices, test/matrices/failing, test/test-benchmark, test/systems)
     into tests/golden/, gzip-compressing anything over 64 KiB;
  2. for every matrix computes y = A x with x_i = 0.25 i (the protocol of
     test/test_spmv.cpp:27-28,45-47) three independent ways -- the C oracle
     (oracle/cask_oracle.c), scipy, and MKL mkl_cspblas_dcsrgemv (the CPU
     library the reference calls, lib/sparse-bench/.../fpgaNaiveCpuCode.cpp:33)
     -- checks they agree under the reference's tolerance, and stores the
     ORACLE's y in spmv_expected.npz;
  3. does not touch known_answers.json, which is transcribed by hand from the
     reference's gtest files (inputs and expected outputs only).

No reference source text is copied: only data files and numbers.
"""
import ctypes
import gzip
import os
import shutil
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
import oracle  # noqa: E402
from oracle import mmio  # noqa: E402

REF = Path("/root/reference/test")
SETS = {"matrices": REF / "matrices", "failing": REF / "matrices" / "failing",
        "benchmark": REF / "test-benchmark", "systems": REF / "systems"}
GZ_OVER = 64 * 1024


def mkl_gemv(csr, x):
    os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
    L = ctypes.CDLL("/opt/conda/lib/libmkl_rt.so.1", mode=ctypes.RTLD_GLOBAL)
    y = np.zeros(csr.n)
    if csr.nnz == 0:
        return y
    tr, nn = ctypes.c_char(b"N"), ctypes.c_int(csr.n)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.mkl_cspblas_dcsrgemv(ctypes.byref(tr), ctypes.byref(nn), p(csr.values), p(csr.row_ptr),
                           p(csr.col_ind), p(x), p(y))
    return y


def main():
    expected = {}
    for label, src in SETS.items():
        dst = HERE / label
        dst.mkdir(exist_ok=True)
        for f in sorted(src.glob("*.mtx")):
            if label == "benchmark" and (SETS["matrices"] / f.name).exists():
                continue                       # byte-identical duplicate of matrices/<name>
            if f.stat().st_size > GZ_OVER:
                out = dst / (f.name + ".gz")
                with open(f, "rb") as fi, gzip.GzipFile(out, "wb", mtime=0) as fo:
                    shutil.copyfileobj(fi, fo)
            else:
                out = dst / f.name
                shutil.copyfile(f, out)
            os.chmod(out, 0o644)
            info = mmio.read_header(out)
            if info.format != "coordinate":
                continue                       # the _b / _sol vectors
            csr = mmio.read_matrix(out)
            x = mmio.test_vector(csr.m)
            y = oracle.csr_spmv(csr.row_ptr, csr.col_ind, csr.values, x)
            A = sp.csr_matrix((csr.values, csr.col_ind, csr.row_ptr), shape=(csr.n, csr.m))
            oracle.assert_almost_equal(A @ x, y, what=f"scipy vs oracle {f.name}")
            if csr.n == csr.m:
                oracle.assert_almost_equal(mkl_gemv(csr, x), y, what=f"MKL vs oracle {f.name}")
            key = f"{label}/{f.name[:-4]}"
            expected[key] = y
            print(f"{key:45s} n={csr.n:6d} nnz={csr.nnz:7d}  |y|_inf={np.abs(y).max() if y.size else 0:.6g}")
    np.savez_compressed(HERE / "spmv_expected.npz", **expected)
    print("wrote", len(expected), "expected vectors")


if __name__ == "__main__":
    main()
