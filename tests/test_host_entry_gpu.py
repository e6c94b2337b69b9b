"""The host-vector entry point cask_hip_spmv (row a6: the function the reference's clients call, Spmv.cpp:185-328) in
every way its vectors can travel (include/cask_hip.h, ABI 7): pageable copies, the staged path (threaded host copies +
a pull kernel + y written into pinned host memory), vectors the caller registered, the engine's registration cache.
Same bits whichever way (the product kernel is the same launch); pointer hygiene: unaligned (8-byte) vectors, vectors
that come back, vectors that are freed and reallocated between calls, x and y that change size with the handle."""
import ctypes
import gc

import numpy as np
import pytest

import oracle
from cask_amd import capi, synth

pytestmark = pytest.mark.gpu


def call(m, x, y):
    rc = capi.load().cask_hip_spmv(m._h, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, capi.load().cask_hip_last_error()


@pytest.fixture()
def restore_mode():
    prev = capi.host_entry_mode("auto")
    yield
    capi.host_entry_mode(prev)


@pytest.mark.parametrize("name", ["cant", "webbase-1M"])
def test_every_mode_gives_the_same_bits(name, restore_mode):
    n, rp, ci, va = synth.small(name, factor=4)                # 125 KB - 2 MB of vectors: the staged path's territory
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    got = {}
    for mode in ("pageable", "staged", "auto", "register_cache"):
        capi.host_entry_mode(mode)
        y = np.full(n, np.nan)
        for _ in range(3):                                      # vectors that come back
            call(m, x, y)
        got[mode] = y
        oracle.assert_almost_equal(y, want, what=f"{name} {mode}")
    capi.host_entry_mode("auto")
    xr, yr = x.copy(), np.full(n, np.nan)
    capi.host_register(xr)
    capi.host_register(yr)
    try:
        call(m, xr, yr)
        xr[:] = 2.0 * x                                         # the CPU writes the registered operand in place, the GPU sees it
        y2 = np.full(n, np.nan)
        call(m, xr, y2)                                         # registered x, ordinary y
        call(m, xr, yr)
    finally:
        capi.host_unregister(xr)
        capi.host_unregister(yr)
    oracle.assert_almost_equal(yr, 2.0 * want, what=f"{name} registered, operand rewritten in place")
    assert np.array_equal(y2, yr)
    for mode in got:
        assert np.array_equal(got[mode], got["pageable"]), mode
    with pytest.raises(ValueError, match="not a registered range"):
        capi.host_unregister(xr)
    m.close()


@pytest.mark.parametrize("mode", ["auto", "staged", "pageable"])
def test_pointer_hygiene(mode, restore_mode):
    """Unaligned vectors (a double* 8 bytes off a 16-byte boundary, at the end of a page), vectors freed and reallocated
    between calls (numpy returns 500 KB arrays to the OS: the next one usually reuses the address with other pages),
    a sliced view in the middle of a bigger array.  The safe modes never hand the caller's memory to the GPU."""
    capi.host_entry_mode(mode)
    n, rp, ci, va = synth.small("cant", factor=2)
    rng = np.random.default_rng(6)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    for off in (0, 1, 3):
        xbuf, ybuf = np.zeros(n + 8), np.full(n + 8, np.nan)
        x, y = xbuf[off:off + n], ybuf[off:off + n]
        x[:] = rng.uniform(-1, 1, n)
        call(m, x, y)
        oracle.assert_almost_equal(y, oracle.csr_spmv(rp, ci, va, x), what=f"offset {off}")
        assert np.all(np.isnan(ybuf[:off])) and np.all(np.isnan(ybuf[off + n:]))      # nothing written outside y
    seen = set()
    for rep in range(12):                                       # freed and reallocated
        x = rng.uniform(-1, 1, n)
        y = np.empty(n)
        seen.add((x.ctypes.data, y.ctypes.data))
        call(m, x, y)
        oracle.assert_almost_equal(y, oracle.csr_spmv(rp, ci, va, x), what=f"fresh vectors, call {rep}")
        del x, y
        gc.collect()
    print("distinct (x, y) address pairs over 12 calls:", len(seen))
    m.close()


def test_registration_cache_with_vectors_that_stay_alive(restore_mode):
    """CASK_HIP_HOST_ENTRY=register_cache (opt-in: the process promises not to unmap a vector it has shown the engine):
    20 vectors alive at once -- more than the cache holds, so entries are evicted and re-registered -- rewritten between
    calls, in rotation; a shorter view of a cached vector; two handles sharing the vectors."""
    capi.host_entry_mode("register_cache")
    n, rp, ci, va = synth.small("atmosmodd", factor=8)
    rng = np.random.default_rng(7)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    m2 = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="merge", tile_width=-1))
    xs = [rng.uniform(-1, 1, n) for _ in range(20)]
    ys = [np.full(n, np.nan) for _ in range(20)]
    for rnd in range(3):
        for i in range(20):
            xs[i] *= -1.5
            call(m if (i + rnd) % 2 else m2, xs[i], ys[i])
        for i in (0, 7, 19):
            oracle.assert_almost_equal(ys[i], oracle.csr_spmv(rp, ci, va, xs[i]), what=f"round {rnd} vector {i}")
    m.close()
    m2.close()
    capi.host_entry_mode("auto")


def test_entry_point_follows_a_replanned_handle(restore_mode):
    """set_params between calls (the staging buffers belong to the handle, the plan changes under them)."""
    n, rp, ci, va = synth.small("G3_circuit", factor=4)
    x = np.random.default_rng(8).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    y = np.empty(n)
    for dp in (dict(variant="merge"), dict(variant="scan"), dict(variant="vector", lanes_per_row=4), dict(variant="slice")):
        m.set_params(capi.make_params(**dp))
        call(m, x, y)
        oracle.assert_almost_equal(y, want, what=str(dp))
    m.close()


@pytest.mark.parametrize("done", ["word", "sync"])
def test_completion_word_and_the_signal_agree(done):
    """How the CPU learns that y is there (cask_hip.hip, r6): a word stored behind the product and polled (the default) or
    hipStreamSynchronize (CASK_HIP_HOST_DONE=sync; read once per process, hence the subprocess).  200 back-to-back calls
    with an operand that changes every call, staged and in place: every result is that call's product."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    code = f"""
import sys, ctypes, numpy as np
sys.path.insert(0, {str(repo)!r})
import oracle
from cask_amd import capi, synth
n, rp, ci, va = synth.small("cant", factor=4)
m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
L = capi.load()
base = oracle.csr_spmv(rp, ci, va, np.ones(n))
for registered in (False, True):
    x, y = np.ones(n), np.zeros(n)
    if registered:
        capi.host_register(x); capi.host_register(y)
    else:
        capi.host_entry_mode("staged")
    for k in range(1, 201):
        x[:] = float(k)
        assert L.cask_hip_spmv(m._h, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p)) == 0
        oracle.assert_almost_equal(y, k * base, what=f"call {{k}} registered={{registered}}")
    if registered:
        capi.host_unregister(x); capi.host_unregister(y)
m.close()
print("ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, CASK_HIP_HOST_DONE=done))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
