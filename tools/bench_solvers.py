#!/usr/bin/env python3
"""BASELINE configs 3 and 5 on one GPU: CG on the G3_circuit-like SPD system and BiCG on the
atmosmodd-like nonsymmetric system (b = A*1), reporting iterations, final residual, device
microseconds per iteration and the algorithmic bytes/iteration of SURVEY.md 8(d):
CG  B_it = B_spmv + 96 n ;  BiCG B_it = 2 B_spmv + 32 n + 120 n."""
import json, sys, time
from pathlib import Path
import numpy as np
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


if __name__ == "__main__":
    only = sys.argv[1:] or ["G3_circuit", "atmosmodd", "cant"]
    for name, solver in (("G3_circuit", "cg"), ("atmosmodd", "bicg"), ("cant", "cg")):
        if name in only:
            run(name, solver, 2000)
    if "pcg" in only or not sys.argv[1:]:
        run_pcg("G3_circuit", "jacobi")
        run_pcg("G3_circuit", "ilu0_mc")
        if "noilu" not in only:
            run_pcg("G3_circuit", "ilu0_unit")
