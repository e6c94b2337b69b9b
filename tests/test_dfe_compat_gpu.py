"""The reference's device function triple on the GPU (include/cask_hip_dfe.h): the test drives it with the
reference host's own stream format and call sequence (oracle/dfe_format.py restates Spmv.cpp) and checks
the result against the golden vectors -- i.e. what running the reference's unchanged Spmv.cpp against
libcask_hip.so would produce, which the reference's mock target cannot (it returns zeros)."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import dfe_format, mmio
from cask_amd import capi
from conftest import golden_matrix_files

pytestmark = pytest.mark.gpu


class Cfg(ctypes.Structure):
    _fields_ = [("num_pipes", ctypes.c_int32), ("num_controllers", ctypes.c_int32), ("cache_size", ctypes.c_int32),
                ("input_width", ctypes.c_int32)]


def make_triple(cfg):
    lib = capi.load()
    p = ctypes.c_void_p
    lib.cask_hip_dfe_dram_write.argtypes = [p, ctypes.c_int64, p, p, p, ctypes.c_char_p]
    lib.cask_hip_dfe_dram_write.restype = None
    lib.cask_hip_dfe_dram_read.argtypes = [p, ctypes.c_int64, p, p, p, ctypes.c_char_p]
    lib.cask_hip_dfe_dram_read.restype = None
    lib.cask_hip_dfe_run.argtypes = [p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64] + [p] * 9
    lib.cask_hip_dfe_run.restype = None
    lib.cask_hip_dfe_reset.restype = None
    ref = ctypes.byref(cfg)
    ptr = lambda a: a.ctypes.data_as(p)  # noqa: E731

    def write(size, sizes, starts, data, routing):
        lib.cask_hip_dfe_dram_write(ref, size, ptr(sizes), ptr(starts), data, routing.encode())

    def read(size, sizes, starts, nbytes, routing):
        buf = ctypes.create_string_buffer(nbytes)
        lib.cask_hip_dfe_dram_read(ref, size, ptr(sizes), ptr(starts), buf, routing.encode())
        return buf.raw

    def run(n_it, n_blocks, vlc, *arrays):
        lib.cask_hip_dfe_run(ref, n_it, n_blocks, vlc, *[ptr(a) for a in arrays])

    return (write, run, read), lib


ARCHS = [  # (num_pipes, num_controllers, cache_size, input_width): shapes of the reference's shipped param files
    (2, 2, 1024, 16), (1, 1, 2048, 8), (6, 3, 256, 3), (4, 1, 4096, 16),
]


@pytest.mark.parametrize("arch", ARCHS, ids=lambda a: "p%d_c%d_cache%d_w%d" % a)
def test_triple_runs_the_reference_stream_format(arch, expected_y):
    pipes, ctrls, cache, width = arch
    cfg = Cfg(pipes, ctrls, cache, width)
    triple, lib = make_triple(cfg)
    for key, path in golden_matrix_files():
        m = mmio.read_matrix(path)
        if m.n > 20000 and arch != ARCHS[0]:
            continue                                            # big ones once: the format is O(n * n_blocks)
        parts = dfe_format.preprocess(m.n, m.m, m.row_ptr, m.col_ind, m.values, pipes, cache, width)
        x = mmio.test_vector(m.m)
        got = dfe_format.spmv_through_triple(triple, m.n, parts, x, pipes, ctrls, cache)
        oracle.assert_almost_equal(got, expected_y[key], what=f"{key} {arch}")
    lib.cask_hip_dfe_reset()


def test_stream_format_restatement_matches_the_cpu_decoder():
    """The numpy format builder against the C decoder of the same format (both restate Spmv.cpp)."""
    m = mmio.read_matrix(dict(golden_matrix_files())["matrices/test_cage6"])
    x = mmio.test_vector(m.m)
    parts = dfe_format.preprocess(m.n, m.m, m.row_ptr, m.col_ind, m.values, 3, 32, 5)
    xp = np.concatenate([x, np.zeros((-x.size) % 32)])
    y = np.concatenate([oracle.partition_decode_spmv(p["n"], p["n_blocks"], 32, 5, False, p["colptr"],
                                                      np.frombuffer(p["records"].tobytes(), dtype=np.uint8), xp)
                        for p in parts])
    oracle.assert_almost_equal(y, oracle.csr_spmv(m.row_ptr, m.col_ind, m.values, x), what="format restatement")


@pytest.mark.parametrize("arch", [(2, 2, 32, 4), (1, 1, 64, 16), (3, 3, 1024, 8)], ids=lambda a: "p%d_c%d_cache%d_w%d" % a)
def test_triple_decodes_run_length_encoded_column_pointers(arch, expected_y):
    """SkipEmptyRowsSpmv's stream (Spmv.hpp:213-250): middle blocks carry runs of empty rows as `length | 1<<31`.
    The reference only ever ran it against the mock; here the GPU decodes it, and the C decoder of the oracle agrees."""
    pipes, ctrls, cache, width = arch
    cfg = Cfg(pipes, ctrls, cache, width)
    triple, lib = make_triple(cfg)
    some_rle = False
    for key, path in golden_matrix_files():
        m = mmio.read_matrix(path)
        if m.n > 16000 or m.m // cache > 600:
            continue
        parts = dfe_format.preprocess(m.n, m.m, m.row_ptr, m.col_ind, m.values, pipes, cache, width, rle=True)
        plain = dfe_format.preprocess(m.n, m.m, m.row_ptr, m.col_ind, m.values, pipes, cache, width)
        some_rle = some_rle or any(p["colptr"].size < q["colptr"].size for p, q in zip(parts, plain))
        x = mmio.test_vector(m.m)
        got = dfe_format.spmv_through_triple(triple, m.n, parts, x, pipes, ctrls, cache)
        oracle.assert_almost_equal(got, expected_y[key], what=f"rle {key} {arch}")
        if m.n <= 200:                                          # and the CPU decoder of the same stream
            xp = np.concatenate([x, np.zeros((-x.size) % cache)])
            y = np.concatenate([oracle.partition_decode_spmv(p["n"], p["n_blocks"], cache, width, True, p["colptr"],
                                                              np.frombuffer(p["records"].tobytes(), dtype=np.uint8), xp)
                                for p in parts])
            oracle.assert_almost_equal(y[: m.n] if m.n >= pipes else y[: m.n], expected_y[key], what=f"rle cpu {key}")
    assert some_rle, "no fixture produced an encoded block: the test would prove nothing"
    lib.cask_hip_dfe_reset()
