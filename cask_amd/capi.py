"""ctypes face of ``libcask_hip.so`` (the C ABI in ``include/cask_hip.h``).

This is plumbing for tests, ``bench.py`` and the multi-GPU driver -- the product
is the shared library.  There is NO CPU fallback: if the library is missing,
or no GPU is visible when a matrix is created, the call raises.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, byref, c_char, c_double, c_int32, c_int64, c_void_p
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libcask_hip.so"

VARIANT_AUTO, VARIANT_VECTOR, VARIANT_MERGE, VARIANT_MERGE_WAVE, VARIANT_SCAN = 0, 1, 2, 3, 4
VARIANT_MERGE_PAIR_REMOVED = 5      # ABI 4-5; rejected since ABI 6 (tests/test_capi_cpu.py, test_spmv_gpu.py)
VARIANT_SLICE = 6                   # ABI 7: short rows row-mapped (lanes_per_row = K), long rows nonzero-mapped, one launch
VARIANT_NAMES = {VARIANT_AUTO: "auto", VARIANT_VECTOR: "vector", VARIANT_MERGE: "merge",
                 VARIANT_MERGE_WAVE: "merge_wave", VARIANT_SCAN: "scan", VARIANT_SLICE: "slice"}

# Every symbol include/cask_hip.h declares (tests check the library exports all of them).
EXPORTED_SYMBOLS = (
    "cask_hip_last_error", "cask_hip_abi_version", "cask_hip_device_count", "cask_hip_device_props_get",
    "cask_hip_csr_create", "cask_hip_csr_create_device", "cask_hip_csr_destroy", "cask_hip_csr_set_params",
    "cask_hip_csr_get_params", "cask_hip_csr_get_info", "cask_hip_spmv", "cask_hip_spmv_device",
    "cask_hip_spmv_dot_device", "cask_hip_spmv_transpose_device", "cask_hip_spmv_time", "cask_hip_tune", "cask_hip_ddot_device",
    "cask_hip_daxpy_device", "cask_hip_daxpby_device", "cask_hip_cg", "cask_hip_bicg",
    "cask_hip_precond_create", "cask_hip_precond_destroy", "cask_hip_precond_factor_values", "cask_hip_precond_info",
    "cask_hip_precond_apply", "cask_hip_precond_apply_device", "cask_hip_trsolve", "cask_hip_pcg",
    "cask_hip_solve_device", "cask_hip_spmv_sequence_device", "cask_hip_spmv_windows_device",
    "cask_hip_device_pci_bus_id", "cask_hip_host_entry_mode", "cask_hip_host_register", "cask_hip_host_unregister",
)
HOST_ENTRY_MODES = {"auto": 0, "pageable": 1, "staged": 2, "register_cache": 3}
SOLVER_CG, SOLVER_BICG = 1, 2
SOLVER_AUTO, SOLVER_COMPOSED, SOLVER_CLASSIC = 0, 1, 2
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, c_int32, c_void_p, c_void_p)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p)


class SolverConfig(Structure):
    _fields_ = [("kind", c_int32), ("mode", c_int32), ("d_shared_base", c_void_p), ("stride", c_int64),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", c_void_p), ("exchange", EXCHANGE_FN),
                ("exchange_user", c_void_p), ("n_full", c_int64)]
PRECOND_JACOBI, PRECOND_ILU0, PRECOND_ILU0_UNIT = 1, 2, 3          # (4, the multicolour ILU(0), was removed in ABI 7)


class Params(Structure):
    _fields_ = [("variant", c_int32), ("lanes_per_row", c_int32), ("tile_width", c_int32),
                ("wg_size", c_int32), ("items_per_thread", c_int32), ("xcd_remap", c_int32),
                ("nontemporal", c_int32), ("index16", c_int32), ("far_columns", c_int32)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_}
        d["variant"] = VARIANT_NAMES.get(d["variant"], d["variant"])
        return d


class CsrInfo(Structure):
    _fields_ = [("n_rows", c_int32), ("n_cols", c_int32), ("nnz", c_int64), ("grid", c_int32),
                ("lds_bytes", c_int32), ("n_long_rows", c_int32), ("n_split_rows", c_int32),
                ("max_row_nnz", c_int32), ("empty_rows", c_int32), ("mean_row_nnz", c_double),
                ("algorithmic_bytes", c_int64), ("fuses_dot", c_int32), ("reserved", c_int32)]


class DeviceProps(Structure):
    _fields_ = [("name", c_char * 64), ("arch", c_char * 32), ("compute_units", c_int32),
                ("lds_bytes_per_cu", c_int32), ("wavefront_size", c_int32), ("clock_mhz", c_int32),
                ("hbm_bytes", c_int64), ("l2_bytes", c_int32), ("reserved", c_int32)]


class TunePoint(Structure):
    _fields_ = [("params", Params), ("usec", c_double), ("gflops", c_double), ("gbytes_per_s", c_double),
                ("valid", c_int32), ("copies", c_int32), ("usec_warm", c_double)]


class CaskHipError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load the engine; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    global LIB_PATH
    if os.environ.get("CASK_HIP_DIAGNOSTIC_LIB"):          # development only: tools/stamps.py, ablation builds
        LIB_PATH = Path(os.environ["CASK_HIP_DIAGNOSTIC_LIB"])
    if not LIB_PATH.exists():
        raise CaskHipError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
                           "the engine has no CPU fallback")
    # torch wheels bundle their own libamdhip64; whichever HIP runtime is loaded first serves the whole
    # process.  Load torch's first, otherwise a later `import torch` finds "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(str(LIB_PATH))
    vp, i32, i64, dbl = c_void_p, c_int32, c_int64, c_double
    L.cask_hip_last_error.restype = ctypes.c_char_p
    L.cask_hip_abi_version.restype = ctypes.c_int
    L.cask_hip_device_count.argtypes = [POINTER(i32)]
    L.cask_hip_device_props_get.argtypes = [i32, POINTER(DeviceProps)]
    L.cask_hip_csr_create.argtypes = [i32, i32, i64, vp, vp, vp, POINTER(Params), POINTER(vp)]
    L.cask_hip_csr_create_device.argtypes = [i32, i32, i64, vp, vp, vp, POINTER(Params), POINTER(vp)]
    L.cask_hip_csr_destroy.argtypes = [vp]
    L.cask_hip_csr_set_params.argtypes = [vp, POINTER(Params)]
    L.cask_hip_csr_get_params.argtypes = [vp, POINTER(Params)]
    L.cask_hip_csr_get_info.argtypes = [vp, POINTER(CsrInfo)]
    L.cask_hip_spmv.argtypes = [vp, vp, vp]
    L.cask_hip_spmv_device.argtypes = [vp, vp, vp, vp]
    if hasattr(L, "cask_hip_spmv_dot_device") or not os.environ.get("CASK_HIP_DIAGNOSTIC_LIB"):
        L.cask_hip_spmv_dot_device.argtypes = [vp, vp, vp, vp, vp, vp]
    L.cask_hip_spmv_transpose_device.argtypes = [vp, vp, vp, vp]
    L.cask_hip_spmv_time.argtypes = [vp, vp, vp, i32, i32, POINTER(dbl), POINTER(dbl)]
    L.cask_hip_tune.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32,
                                POINTER(TunePoint), i32, POINTER(i32), POINTER(i32)]
    L.cask_hip_ddot_device.argtypes = [i64, vp, vp, vp, vp]
    L.cask_hip_daxpy_device.argtypes = [i64, dbl, vp, vp, vp, vp, vp]
    L.cask_hip_daxpby_device.argtypes = [i64, dbl, vp, dbl, dbl, vp, vp, vp, vp]
    for name in ("cask_hip_cg", "cask_hip_bicg"):
        getattr(L, name).argtypes = [vp, vp, vp, i32, dbl, POINTER(i32), POINTER(i32), POINTER(dbl)]
    if hasattr(L, "cask_hip_pcg") or not os.environ.get("CASK_HIP_DIAGNOSTIC_LIB"):
        L.cask_hip_precond_create.argtypes = [i32, i32, i64, vp, vp, vp, POINTER(vp)]
        L.cask_hip_precond_destroy.argtypes = [vp]
        L.cask_hip_precond_factor_values.argtypes = [vp, vp]
        L.cask_hip_precond_info.argtypes = [vp, POINTER(i32), POINTER(i32), POINTER(i32)]
        L.cask_hip_precond_apply.argtypes = [vp, vp, vp]
        L.cask_hip_precond_apply_device.argtypes = [vp, vp, vp, vp]
        L.cask_hip_trsolve.argtypes = [i32, i64, vp, vp, vp, i32, vp, vp]
        L.cask_hip_pcg.argtypes = [vp, vp, vp, vp, i32, dbl, POINTER(i32), POINTER(i32), POINTER(dbl)]
    if hasattr(L, "cask_hip_spmv_sequence_device"):
        L.cask_hip_spmv_sequence_device.argtypes = [vp, i32, vp, vp, i32, vp]
    if hasattr(L, "cask_hip_spmv_windows_device"):
        L.cask_hip_spmv_windows_device.argtypes = [vp, i32, vp, vp, i32, i32, i32, vp, vp]
    if hasattr(L, "cask_hip_solve_device"):
        L.cask_hip_solve_device.argtypes = [vp, vp, POINTER(SolverConfig), vp, vp, i32, dbl, POINTER(i32), POINTER(i32),
                                            POINTER(dbl), vp]
    for name in EXPORTED_SYMBOLS:
        if os.environ.get("CASK_HIP_DIAGNOSTIC_LIB") and not hasattr(L, name):
            continue                                            # an older build loaded for an A/B timing
        f = getattr(L, name)
        if name not in ("cask_hip_last_error",):
            f.restype = ctypes.c_int
    _lib = L
    return L


def _check(rc: int):
    if rc == 0:
        return
    msg = load().cask_hip_last_error().decode(errors="replace")
    if rc == 1:                      # CASK_HIP_ERR_INVALID <-> std::invalid_argument
        raise ValueError(msg)
    raise CaskHipError(f"cask_hip error {rc}: {msg}")


def device_count() -> int:
    n = c_int32(0)
    _check(load().cask_hip_device_count(byref(n)))
    return n.value


def device_props(device: int = 0) -> dict:
    p = DeviceProps()
    _check(load().cask_hip_device_props_get(device, byref(p)))
    return {"name": p.name.decode(), "arch": p.arch.decode(), "compute_units": p.compute_units,
            "lds_bytes_per_cu": p.lds_bytes_per_cu, "wavefront_size": p.wavefront_size,
            "clock_mhz": p.clock_mhz, "hbm_bytes": p.hbm_bytes, "l2_bytes": p.l2_bytes}


def device_pci_bus_id(device: int = 0) -> str:
    L = load()
    L.cask_hip_device_pci_bus_id.argtypes = [c_int32, c_void_p, c_int32]
    buf = ctypes.create_string_buffer(64)
    _check(L.cask_hip_device_pci_bus_id(device, buf, 64))
    return buf.value.decode()


def make_params(variant=0, lanes_per_row=0, tile_width=0, wg_size=0, items_per_thread=0,
                xcd_remap=0, nontemporal=0, index16=0, far_columns=0) -> Params:
    if isinstance(variant, str):
        variant = {v: k for k, v in VARIANT_NAMES.items()}[variant]
    return Params(variant, lanes_per_row, tile_width, wg_size, items_per_thread, xcd_remap, nontemporal, index16,
                  far_columns)


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(c_void_p) if a.size else None


def _int_array(vals):
    if vals is None:
        return None, 0
    a = np.ascontiguousarray(list(vals), dtype=np.int32)
    return a, a.size


class CsrMatrix:
    """A device-resident CSR matrix + launch plan (``cask_hip_matrix*``)."""

    def __init__(self, handle, n_rows, n_cols, nnz, keep=()):
        self._h = handle
        self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, nnz
        self._keep = keep            # device tensors borrowed by the handle

    @classmethod
    def from_host(cls, n_rows, n_cols, row_ptr, col_ind, values, params: Params | None = None):
        rp, ci, va = _np(row_ptr, np.int32), _np(col_ind, np.int32), _np(values, np.float64)
        if rp.size != n_rows + 1:
            raise ValueError("row_ptr must have n_rows+1 entries")
        if ci.size != va.size:
            raise ValueError("col_ind and values differ in length")
        h = c_void_p()
        _check(load().cask_hip_csr_create(n_rows, n_cols, ci.size, _p(rp), _p(ci), _p(va),
                                          byref(params) if params is not None else None, byref(h)))
        return cls(h, n_rows, n_cols, int(ci.size))

    @classmethod
    def from_device(cls, n_rows, n_cols, row_ptr_t, col_ind_t, values_t, params: Params | None = None):
        """torch CUDA tensors (int32, int32, float64); borrowed, kept alive by this object."""
        h = c_void_p()
        nnz = int(col_ind_t.numel())
        _check(load().cask_hip_csr_create_device(
            n_rows, n_cols, nnz, c_void_p(row_ptr_t.data_ptr()),
            c_void_p(col_ind_t.data_ptr()) if nnz else None,
            c_void_p(values_t.data_ptr()) if nnz else None,
            byref(params) if params is not None else None, byref(h)))
        return cls(h, n_rows, n_cols, nnz, keep=(row_ptr_t, col_ind_t, values_t))

    def close(self):
        if self._h:
            load().cask_hip_csr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- design point ---------------------------------------------------------
    def set_params(self, params: Params | None = None, **kw):
        if params is None:
            params = make_params(**kw)
        _check(load().cask_hip_csr_set_params(self._h, byref(params)))

    @property
    def params(self) -> Params:
        p = Params()
        _check(load().cask_hip_csr_get_params(self._h, byref(p)))
        return p

    @property
    def info(self) -> CsrInfo:
        i = CsrInfo()
        _check(load().cask_hip_csr_get_info(self._h, byref(i)))
        return i

    # -- products ---------------------------------------------------------------
    def spmv(self, x) -> np.ndarray:
        """Host vectors (the Spmv::spmv contract, Spmv.cpp:185-328)."""
        x = _np(x, np.float64)
        if x.size != self.n_cols:
            raise ValueError(f"x has {x.size} entries, matrix has {self.n_cols} columns")
        y = np.empty(self.n_rows, dtype=np.float64)
        _check(load().cask_hip_spmv(self._h, _p(x), _p(y)))
        return y

    def spmv_device(self, x_t, y_t, stream=None):
        """torch CUDA tensors; launches on ``stream`` (default: torch's current stream)."""
        _check(load().cask_hip_spmv_device(self._h, c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()),
                                           c_void_p(_stream_ptr(stream))))

    def spmv_dot_device(self, x_t, y_t, w_t, out_t, stream=None):
        """y = A x and out[0] = w . y in one pass (MERGE plans: fused epilogue)."""
        _check(load().cask_hip_spmv_dot_device(self._h, c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()),
                                               c_void_p(w_t.data_ptr()), c_void_p(out_t.data_ptr()),
                                               c_void_p(_stream_ptr(stream))))

    def set_halo_sources(self, n_own, addr_t):
        """Columns >= n_own are read from the device addresses in ``addr_t`` (int64 CUDA tensor) by the
        product kernel itself (include/cask_hip_p2p.h); ``addr_t=None`` restores the plain layout."""
        L = load()
        L.cask_hip_csr_set_halo_sources.argtypes = [c_void_p, c_int32, c_void_p]
        L.cask_hip_csr_set_halo_sources.restype = ctypes.c_int
        _check(L.cask_hip_csr_set_halo_sources(self._h, int(n_own),
                                               c_void_p(addr_t.data_ptr()) if addr_t is not None else None))
        self._halo_keep = addr_t

    def spmv_transpose_device(self, x_t, y_t, stream=None):
        _check(load().cask_hip_spmv_transpose_device(self._h, c_void_p(x_t.data_ptr()),
                                                     c_void_p(y_t.data_ptr()), c_void_p(_stream_ptr(stream))))

    def time(self, x_t, y_t, warmup=5, iters=50):
        med, mn = c_double(0), c_double(0)
        _check(load().cask_hip_spmv_time(self._h, c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()),
                                         warmup, iters, byref(med), byref(mn)))
        return med.value, mn.value

    def tune(self, variants=None, lanes=None, tiles=None, wg_sizes=None, items=None, warmup=1, iters=0,
             max_results=4096):
        """Measured DSE; leaves the best point active. Returns (points, best_index)."""
        arrs = [_int_array(v) for v in (variants, lanes, tiles, wg_sizes, items)]
        res = (TunePoint * max_results)()
        n, best = c_int32(0), c_int32(-1)
        args = []
        for a, cnt in arrs:
            args += [_p(a) if a is not None else None, cnt]
        _check(load().cask_hip_tune(self._h, *args, warmup, iters, res, max_results, byref(n), byref(best)))
        pts = [{"params": res[i].params.as_dict(), "usec": res[i].usec, "usec_warm": res[i].usec_warm,
                "copies": res[i].copies, "gflops": res[i].gflops,
                "gbytes_per_s": res[i].gbytes_per_s, "valid": bool(res[i].valid)} for i in range(n.value)]
        return pts, best.value

    # -- solvers ----------------------------------------------------------------
    def _solve(self, fn, rhs, x0, maxiters, tol):
        rhs = _np(rhs, np.float64)
        x = np.zeros(self.n_rows) if x0 is None else _np(x0, np.float64).copy()
        if rhs.size != self.n_rows or x.size != self.n_rows:
            raise ValueError("rhs/x0 size mismatch")
        it, conv, us = c_int32(0), c_int32(0), c_double(0)
        _check(fn(self._h, _p(rhs), _p(x), int(maxiters), float(tol), byref(it), byref(conv), byref(us)))
        return x, it.value, bool(conv.value), us.value

    def cg(self, rhs, x0=None, maxiters=2000, tol=1e-5):
        """(x, iterations, converged, usec_per_iteration); pcg semantics (SparseLinearSolvers.hpp:162-239)."""
        return self._solve(load().cask_hip_cg, rhs, x0, maxiters, tol)

    def bicg(self, rhs, x0=None, maxiters=2000, tol=1e-5):
        return self._solve(load().cask_hip_bicg, rhs, x0, maxiters, tol)

    def solve_device(self, b_t, x_t, kind="cg", transposed=None, mode=SOLVER_AUTO, maxiters=2000, tol=1e-5,
                     shared_base=0, stride=0, allreduce=None, exchange=None, n_full=0, stream=None):
        """CG / BiCG on device vectors (``cask_hip_solve_device``); ``x_t`` holds the initial guess and receives
        the solution.  ``allreduce(ptr, count, stream) -> int`` / ``exchange(local_ptr, full_ptr, stream) -> int``
        are python callables for the row-sharded forms (see include/cask_hip.h).  Returns
        (iterations, converged, usec_per_iteration)."""
        cfg = SolverConfig()
        cfg.kind = {"cg": SOLVER_CG, "bicg": SOLVER_BICG}.get(kind, kind)
        cfg.mode = mode
        cfg.d_shared_base = shared_base or None
        cfg.stride = stride
        cfg.n_full = n_full
        keep, raised = [], []

        def guard(fn):
            # an exception inside a ctypes callback is printed and swallowed, and C would see 0 = success: the solve
            # would go on with un-reduced scalars.  Report failure to the engine and re-raise after the call.
            def wrapped(*a):
                try:
                    return int(fn(*a) or 0)
                except BaseException as e:  # noqa: BLE001
                    raised.append(e)
                    return 1
            return wrapped
        if isinstance(allreduce, NativeComm):                  # RCCL issued by the engine itself (include/cask_hip_rccl.h)
            cfg.allreduce = ctypes.cast(load().cask_hip_rccl_allreduce, ALLREDUCE_FN)
            cfg.allreduce_user = allreduce.handle
        elif type(allreduce).__name__ == "PushExchange":       # peer-store reduction (include/cask_hip_p2p.h): no library
            cfg.allreduce = ctypes.cast(load().cask_hip_push_allreduce, ALLREDUCE_FN)
            cfg.allreduce_user = allreduce.handle
        elif allreduce is not None:
            cb = ALLREDUCE_FN(guard(lambda p, c, s, u: allreduce(p, c, s)))
            cfg.allreduce = cb
            keep.append(cb)
        if isinstance(exchange, NativeComm):
            cfg.exchange = ctypes.cast(load().cask_hip_rccl_allgather, EXCHANGE_FN)
            cfg.exchange_user = exchange.handle
        elif exchange is not None:
            cb2 = EXCHANGE_FN(guard(lambda a, b, s, u: exchange(a, b, s)))
            cfg.exchange = cb2
            keep.append(cb2)
        it, conv, us = c_int32(0), c_int32(0), c_double(0)
        rc = load().cask_hip_solve_device(self._h, transposed._h if transposed is not None else None, byref(cfg),
                                          c_void_p(b_t.data_ptr()), c_void_p(x_t.data_ptr()), int(maxiters), float(tol),
                                          byref(it), byref(conv), byref(us), c_void_p(_stream_ptr(stream)))
        del keep
        if raised:
            raise raised[0]
        _check(rc)
        return it.value, bool(conv.value), us.value

    def pcg(self, precond, rhs, x0=None, maxiters=2000, tol=1e-5):
        """Preconditioned CG (pcg<double, Precon>, SparseLinearSolvers.hpp:162-239) with a ``Preconditioner``."""
        fn = load().cask_hip_pcg
        return self._solve(lambda h, *a: fn(h, precond._h if precond is not None else None, *a), rhs, x0, maxiters, tol)


RCCL_SYMBOLS = ("cask_hip_rccl_unique_id", "cask_hip_rccl_comm_create", "cask_hip_rccl_comm_destroy",
                "cask_hip_rccl_allreduce", "cask_hip_rccl_allgather", "cask_hip_rccl_comm_set_stride",
                "cask_hip_rccl_comm_info")


class NativeComm:
    """An RCCL communicator owned by the engine (include/cask_hip_rccl.h): the all-reduce / operand all-gather of a
    row-sharded solve issued from C on the solver's stream.  ``unique_id()`` on rank 0, ship the bytes, then
    ``NativeComm(id, rank, world, bounds)`` on every rank (collective)."""

    @staticmethod
    def _lib():
        L = load()
        if not getattr(L, "_rccl_bound", False):
            L.cask_hip_rccl_unique_id.argtypes = [c_void_p]
            L.cask_hip_rccl_comm_create.argtypes = [c_void_p, c_int32, c_int32, c_void_p, POINTER(c_void_p)]
            L.cask_hip_rccl_comm_destroy.argtypes = [c_void_p]
            L.cask_hip_rccl_allreduce.argtypes = [c_void_p, c_int32, c_void_p, c_void_p]
            L.cask_hip_rccl_allgather.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
            L.cask_hip_rccl_comm_set_stride.argtypes = [c_void_p, c_int64]
            L.cask_hip_rccl_comm_info.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), c_void_p, c_int32]
            for name in RCCL_SYMBOLS:
                getattr(L, name).restype = ctypes.c_int
            L._rccl_bound = True
        return L

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_ubyte * 128)()
        _check(NativeComm._lib().cask_hip_rccl_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes, rank: int, world: int, bounds=None, stride: int = 0):
        """``stride`` > 0: the gathered vector is laid out with that padded stride (rank g's slice at g*stride) and
        the all-gather is ONE ncclAllGather of ``stride`` doubles per rank whatever the row partition."""
        L = self._lib()
        b = np.ascontiguousarray(bounds, dtype=np.int64) if bounds is not None else None
        h = c_void_p()
        idbuf = (ctypes.c_ubyte * 128).from_buffer_copy(uid)
        _check(L.cask_hip_rccl_comm_create(idbuf, rank, world, _p(b) if b is not None else None, byref(h)))
        self.handle = h
        if stride:
            _check(L.cask_hip_rccl_comm_set_stride(h, int(stride)))

    def info(self) -> dict:
        """What RCCL reports for this communicator: ranks, this rank, its device and that device's PCI bus id."""
        n, r, d = c_int32(-1), c_int32(-1), c_int32(-1)
        pci = ctypes.create_string_buffer(64)
        _check(self._lib().cask_hip_rccl_comm_info(self.handle, byref(n), byref(r), byref(d), pci, 64))
        return {"comm_nranks": n.value, "comm_rank": r.value, "device": d.value, "pci_bus_id": pci.value.decode()}

    def allreduce(self, t, stream=None):
        _check(self._lib().cask_hip_rccl_allreduce(c_void_p(t.data_ptr()), t.numel(), c_void_p(_stream_ptr(stream)), self.handle))

    def allgather(self, local_t, full_t, stream=None):
        _check(self._lib().cask_hip_rccl_allgather(c_void_p(local_t.data_ptr()), c_void_p(full_t.data_ptr()),
                                                   c_void_p(_stream_ptr(stream)), self.handle))

    def close(self):
        if self.handle:
            self._lib().cask_hip_rccl_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Preconditioner:
    """Jacobi or ILU(0) (``cask_hip_precond*``): factored once on the host, applied on the device."""

    def __init__(self, kind, n, row_ptr, col_ind, values):
        kind = {"jacobi": PRECOND_JACOBI, "ilu0": PRECOND_ILU0, "ilu0_unit": PRECOND_ILU0_UNIT}.get(kind, kind)
        rp, ci, va = _np(row_ptr, np.int32), _np(col_ind, np.int32), _np(values, np.float64)
        if rp.size != n + 1 or ci.size != va.size:
            raise ValueError("malformed CSR arrays")
        self.n, self.nnz, self.kind = n, int(ci.size), kind
        h = c_void_p()
        _check(load().cask_hip_precond_create(kind, n, ci.size, _p(rp), _p(ci), _p(va), byref(h)))
        self._h = h

    def factor_values(self) -> np.ndarray:
        out = np.empty(self.nnz, dtype=np.float64)
        _check(load().cask_hip_precond_factor_values(self._h, _p(out) if out.size else None))
        return out

    def info(self):
        a, b, c = c_int32(0), c_int32(0), c_int32(0)
        _check(load().cask_hip_precond_info(self._h, byref(a), byref(b), byref(c)))
        return {"levels_lower": a.value, "levels_upper": b.value, "launches_per_apply": c.value}

    def apply(self, r) -> np.ndarray:
        r = _np(r, np.float64)
        if r.size != self.n:
            raise ValueError("vector length mismatch")
        z = np.empty(self.n, dtype=np.float64)
        _check(load().cask_hip_precond_apply(self._h, _p(r), _p(z)))
        return z

    def apply_device(self, r_t, z_t, stream=None):
        _check(load().cask_hip_precond_apply_device(self._h, c_void_p(r_t.data_ptr()), c_void_p(z_t.data_ptr()),
                                                    c_void_p(_stream_ptr(stream))))

    def close(self):
        if self._h:
            load().cask_hip_precond_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_entry_mode(mode) -> str:
    """Process-wide choice of how ``cask_hip_spmv`` moves host vectors (include/cask_hip.h); returns the previous one."""
    names = {v: k for k, v in HOST_ENTRY_MODES.items()}
    return names[load().cask_hip_host_entry_mode(HOST_ENTRY_MODES[mode] if isinstance(mode, str) else int(mode))]


def host_register(a: np.ndarray):
    """Declare a numpy array long-lived host memory the GPU may access in place (it must stay alive until
    ``host_unregister``)."""
    L = load()
    L.cask_hip_host_register.argtypes = [c_void_p, ctypes.c_size_t]
    _check(L.cask_hip_host_register(c_void_p(a.ctypes.data), a.nbytes))


def host_unregister(a: np.ndarray):
    L = load()
    L.cask_hip_host_unregister.argtypes = [c_void_p]
    _check(L.cask_hip_host_unregister(c_void_p(a.ctypes.data)))


def trsolve(n, row_ptr, col_ind, values, rhs, lower=True) -> np.ndarray:
    """cask::mkl::unittrsolve on the device: the chosen triangle with its stored diagonal."""
    rp, ci, va, b = _np(row_ptr, np.int32), _np(col_ind, np.int32), _np(values, np.float64), _np(rhs, np.float64)
    x = np.empty(n, dtype=np.float64)
    _check(load().cask_hip_trsolve(n, ci.size, _p(rp), _p(ci), _p(va), int(bool(lower)), _p(b), _p(x)))
    return x


def spmv_sequence_device(mats, x_t, y_t, k, stream=None):
    """k products in stream order from one call: product i uses mats[i % len(mats)] (cask_hip_spmv_sequence_device)."""
    arr = (c_void_p * len(mats))(*[m._h for m in mats])
    _check(load().cask_hip_spmv_sequence_device(arr, len(mats), c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()),
                                                int(k), c_void_p(_stream_ptr(stream))))


def spmv_windows_device(mats, x_t, y_t, k, windows, as_graph=False, stream=None):
    """`windows` timed windows of k products each from one call (cask_hip_spmv_windows_device); returns after the last
    one has completed with the device microseconds of every window.  ``as_graph``: the launches are the kernel nodes of
    one graph with event-record nodes at the window boundaries (raises CaskHipError where the runtime cannot do that)."""
    arr = (c_void_p * len(mats))(*[m._h for m in mats])
    usec = np.zeros(int(windows), dtype=np.float64)
    _check(load().cask_hip_spmv_windows_device(arr, len(mats), c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()),
                                               int(k), int(windows), int(bool(as_graph)), _p(usec),
                                               c_void_p(_stream_ptr(stream))))
    return usec


def _stream_ptr(stream):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


def ddot_device(x_t, y_t, out_t, stream=None):
    _check(load().cask_hip_ddot_device(x_t.numel(), c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()),
                                       c_void_p(out_t.data_ptr()), c_void_p(_stream_ptr(stream))))


def daxpy_device(x_t, y_t, sign=1.0, num_t=None, den_t=None, stream=None):
    """y += sign * (num/den) * x with device-resident scalars."""
    _check(load().cask_hip_daxpy_device(
        x_t.numel(), float(sign), c_void_p(num_t.data_ptr()) if num_t is not None else None,
        c_void_p(den_t.data_ptr()) if den_t is not None else None,
        c_void_p(x_t.data_ptr()), c_void_p(y_t.data_ptr()), c_void_p(_stream_ptr(stream))))


def daxpby_device(alpha, x_t, y_t, beta=0.0, sign=1.0, num_t=None, den_t=None, stream=None):
    """y = alpha*x + b*y, b = beta or sign*(num/den)."""
    _check(load().cask_hip_daxpby_device(
        x_t.numel(), float(alpha), c_void_p(x_t.data_ptr()), float(sign), float(beta),
        c_void_p(num_t.data_ptr()) if num_t is not None else None,
        c_void_p(den_t.data_ptr()) if den_t is not None else None,
        c_void_p(y_t.data_ptr()), c_void_p(_stream_ptr(stream))))
