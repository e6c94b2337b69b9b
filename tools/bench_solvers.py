#!/usr/bin/env python3
"""BASELINE configs 3 and 5 on one GPU: CG on the G3_circuit-like SPD system and BiCG on the
atmosmodd-like nonsymmetric system (b = A*1), reporting iterations, final residual, device
microseconds per iteration and the algorithmic bytes/iteration of SURVEY.md 8(d):
CG  B_it = B_spmv + 96 n ;  BiCG B_it = 2 B_spmv + 32 n + 120 n."""
import ctypes, json, os, sys, time
from pathlib import Path
import numpy as np
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")          # (before MKL loads: pinned GNU-OpenMP teams, as bench.py)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cask_amd import capi, synth

def run(name, solver, maxiters):
    n, rp, ci, va, src = synth.load_or_make(name)
    import os
    prm = capi.make_params(nontemporal=int(os.environ.get("CASK_SOLVER_NT", "0")))   # 0 = default (streaming), -1 = cached
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, prm)
    ones = np.random.default_rng(5).uniform(-1, 1, n)         # the solution the solver must find
    b = m.spmv(ones)
    t0 = time.perf_counter()
    x, it, conv, us = (m.cg if solver == "cg" else m.bicg)(b, maxiters=maxiters)
    wall = time.perf_counter() - t0
    res = float(np.linalg.norm(b - m.spmv(x)))
    b_spmv = synth.algorithmic_bytes(n, n, ci.size)
    b_it = b_spmv + 96 * n if solver == "cg" else 2 * b_spmv + 152 * n
    f_it = 2 * ci.size + 12 * n if solver == "cg" else 4 * ci.size + 20 * n
    out = {"matrix": name, "solver": solver, "n": n, "nnz": int(ci.size), "iterations": it, "converged": conv,
           "residual_2norm": res, "usec_per_iteration": round(us, 2), "wall_s": round(wall, 3),
           "algorithmic_bytes_per_iteration": b_it, "gbs_algorithmic": round(b_it / us * 1e-3, 1),
           "pct_hbm_peak": round(100 * b_it / us * 1e-3 / 8000, 1), "gflops": round(f_it / us * 1e-3, 1),
           "design_point": m.params.as_dict(), "max_err_vs_x0": float(np.abs(x - ones).max())}
    print(json.dumps(out))
    m.close()

def run_pcg(name, kind, maxiters=2000):
    """Preconditioned CG: iterations and device time per pass, and what the triangular solves cost."""
    n, rp, ci, va, src = synth.load_or_make(name)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = m.spmv(x0)
    t0 = time.perf_counter()
    pc = capi.Preconditioner(kind, n, rp, ci, va)
    setup = time.perf_counter() - t0
    x, it, conv, us = m.pcg(pc, b, maxiters=maxiters)
    out = {"matrix": name, "solver": "pcg", "preconditioner": kind, "n": n, "nnz": int(ci.size), "iterations": it,
           "converged": conv, "usec_per_iteration": round(us, 2), "setup_s": round(setup, 3),
           "residual_2norm": float(np.linalg.norm(b - m.spmv(x))), "max_err_vs_x0": float(np.abs(x - x0).max()),
           **pc.info()}
    print(json.dumps(out))
    pc.close()
    m.close()


def _median_time(fn, seconds=1.0, max_calls=200):
    """Median seconds per call of fn(), after two warm-up calls, over at most `seconds` / `max_calls`."""
    fn(); fn()
    ts, t_all = [], time.perf_counter()
    while len(ts) < max_calls and (time.perf_counter() - t_all < seconds or len(ts) < 3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def run_trsv_compare(name, gpu_pcg_passes=24):
    """Row f4 next to the reference's CPU path: ONE ILU(0) application -- the two triangular solves
    ILUPreconditioner::apply hands to mkl_dcsrtrsv (src/runtime/SparseLinearSolvers.hpp:143-151, MklLayer.hpp:29-85:
    uplo 'l' then 'u', trans 'N', diag 'N', 1-based CSR of the extracted triangles) -- on MKL with 1 and N pinned threads
    and on the GPU (the engine's default schedule, k_trsv_walk2), on the SAME factors; then whole passes: pcg with the
    ILU(0) preconditioner against pcg without one (Identity), CPU (MKL calls of the reference's recurrence) and GPU."""
    import torch
    import bench
    n, rp, ci, va, src = synth.load_or_make(name)
    out = {"matrix": name, "n": n, "nnz": int(ci.size), "what": "one ILU(0) application = two triangular solves; pcg passes"}
    t0 = time.perf_counter()
    pc = capi.Preconditioner("ilu0_unit", n, rp, ci, va)       # the textbook application (unit lower diagonal): the one that solves
    out["factor_and_plan_seconds"] = round(time.perf_counter() - t0, 2)
    out.update(pc.info())
    f = pc.factor_values()                                     # factored values in the pattern of the input (ILUPreconditioner::pc)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    tri = {}
    for key, keep in (("l", ci <= rows), ("u", ci >= rows)):
        trp = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows[keep], minlength=n), out=trp[1:])
        tri[key] = (trp + 1, (ci[keep] + 1).astype(np.int32), np.ascontiguousarray(f[keep]))
    nnz_l, nnz_u = int(tri["l"][1].size), int(tri["u"][1].size)
    # algorithmic bytes of one application: per solve 12 B per stored entry (value + column) + per row its row pointer,
    # right-hand side and result (4 + 8 + 8)
    alg_bytes = 12 * (nnz_l + nnz_u) + 2 * 20 * n
    out["algorithmic_bytes_per_application"] = alg_bytes
    r = np.random.default_rng(1).standard_normal(n)
    # ---- GPU
    rt = torch.from_numpy(r).cuda()
    zt = torch.zeros_like(rt)
    pc.apply_device(rt, zt)
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pc.apply_device(rt, zt)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    z_gpu = zt.cpu().numpy()
    out["gpu"] = {"ms_per_application": round(min(times), 3), "schedule": os.environ.get("CASK_HIP_TRSV", "default: lanes for runs of long rows, walk2 otherwise"),
                  "gbs_algorithmic": round(alg_bytes / (min(times) * 1e-3) / 1e9, 2)}
    # ---- CPU: mkl_dcsrtrsv as the reference calls it
    mkl = bench.load_mkl()
    if mkl is None:
        out["cpu"] = {"error": "no MKL on this host"}
    else:
        p_ = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        nn, trN = ctypes.c_int(n), ctypes.c_char(b"N")
        y, z = np.zeros(n), np.zeros(n)

        def apply(diag_l):
            lo, up, dl, du = ctypes.c_char(b"l"), ctypes.c_char(b"u"), ctypes.c_char(diag_l), ctypes.c_char(b"N")
            a, ia, ja = tri["l"][2], tri["l"][0], tri["l"][1]
            mkl.mkl_dcsrtrsv(ctypes.byref(lo), ctypes.byref(trN), ctypes.byref(dl), ctypes.byref(nn), p_(a), p_(ia), p_(ja), p_(r), p_(y))
            a, ia, ja = tri["u"][2], tri["u"][0], tri["u"][1]
            mkl.mkl_dcsrtrsv(ctypes.byref(up), ctypes.byref(trN), ctypes.byref(du), ctypes.byref(nn), p_(a), p_(ia), p_(ja), p_(y), p_(z))
        mkl.MKL_Get_Max_Threads.restype = ctypes.c_int
        max_threads = int(mkl.MKL_Get_Max_Threads())
        cpu = {"host_cores": os.cpu_count(), "routine": "mkl_dcsrtrsv('l','N',diag) then ('u','N','N'), 1-based triangles with the diagonal "
                                                          "(MklLayer.hpp:29-85 as ILUPreconditioner::apply calls it)"}
        for t in sorted({1, min(16, max_threads), max_threads}):
            mkl.MKL_Set_Num_Threads(ctypes.c_int(t))
            for label, diag in (("unit_lower", b"U"), ("reference_diag_N", b"N")):
                sec, calls = _median_time(lambda: apply(diag), seconds=1.5)
                cpu.setdefault(label, {})[str(t)] = {"ms_per_application": round(sec * 1e3, 3), "calls": calls,
                                                     "gbs_algorithmic": round(alg_bytes / sec / 1e9, 2)}
        apply(b"U")
        scale = max(1.0, float(np.abs(z).max()))
        cpu["max_abs_diff_gpu_vs_mkl_unit_lower"] = float(np.abs(z - z_gpu).max() / scale)
        out["cpu"] = cpu
        best_cpu = min(v["ms_per_application"] for v in cpu["unit_lower"].values())
        out["faster"] = "cpu (MKL)" if best_cpu < out["gpu"]["ms_per_application"] else "gpu"
        out["gpu_over_best_cpu"] = round(out["gpu"]["ms_per_application"] / best_cpu, 2)
        # ---- whole passes on the CPU: the reference's pcg recurrence on MKL calls, Identity and ILU
        t16 = min(16, max_threads)
        b = np.asarray(r)
        cg_s = bench.mkl_solver_passes(mkl, "cg", rp, ci, va, b, 10, t16) / 10
        mkl.MKL_Set_Num_Threads(ctypes.c_int(t16))
        ilu_s, _ = _median_time(lambda: apply(b"U"), seconds=1.0)
        out["cpu_passes"] = {"threads": t16, "pcg_identity_ms_per_pass": round(cg_s * 1e3, 3),
                             "pcg_ilu0_ms_per_pass": round((cg_s + ilu_s) * 1e3, 3),
                             "note": "pcg<Identity> = product (mkl_cspblas_dcsrgemv) + ddot/daxpy/daxpby; pcg<ILU> adds one application "
                                     "and one more ddot (r.z, counted with the Identity pass's r.r)"}
    # ---- whole passes on the GPU
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    bb = m.spmv(x0)
    _, it_cg, conv_cg, us_cg = m.cg(bb, maxiters=200)
    _, it_p, conv_p, us_p = m.pcg(pc, bb, maxiters=gpu_pcg_passes)
    out["gpu_passes"] = {"cg_identity_usec_per_pass": round(us_cg, 2), "pcg_ilu0_unit_usec_per_pass": round(us_p, 2),
                         "pcg_passes_timed": it_p + 1}
    print(json.dumps(out))
    pc.close()
    m.close()


if __name__ == "__main__":
    if sys.argv[1:2] == ["trsv"]:                            # python tools/bench_solvers.py trsv [matrix ...]
        for name in sys.argv[2:] or ["G3_circuit", "cant", "atmosmodd"]:
            run_trsv_compare(name)
        sys.exit(0)
    only = sys.argv[1:] or ["G3_circuit", "atmosmodd", "cant"]
    for name, solver in (("G3_circuit", "cg"), ("atmosmodd", "bicg"), ("cant", "cg")):
        if name in only:
            run(name, solver, 2000)
    if "pcg" in only or not sys.argv[1:]:
        run_pcg("G3_circuit", "jacobi")
        if "noilu" not in only:
            run_pcg("G3_circuit", "ilu0_unit")
