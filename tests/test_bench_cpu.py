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


# ---- round 4: the windowed clock and the first-contact self-checks (host logic, CPU tensors) -------------------------
def test_window_count_and_spread_fields():
    """the driver's `--steps 20` line times >= 31 windows -- as many as it takes to reach 2000 steps,
    at most 101 -- the same number on every rank (a function of the arguments alone), and reports their spread."""
    a = bench.parse_args(["--steps", "20"])
    assert bench.n_windows(a) == 101
    assert bench.n_windows(bench.parse_args(["--steps", "1000"])) == 31
    assert bench.n_windows(bench.parse_args(["--steps", "20", "--windows", "35"])) == 35
    ms = [0.170 + 0.001 * i for i in range(31)]
    tw = {"windows": 31, "ms": ms, "median": float(np.median(ms)), "min": min(ms), "p10": ms[3], "p90": ms[27], "max": max(ms),
          "mean": float(np.mean(ms)), "mean_max": float(np.mean(ms)), "first": ms[0]}
    f = bench.window_fields(tw, 20)
    assert f["windows"] == 31 and f["ms_per_step_min"] <= f["ms_per_step_p10"] <= f["ms_per_step_p90"] <= f["ms_per_step_max"]
    assert abs(f["window_spread_pct"] - 100 * (ms[27] - ms[3]) / tw["median"]) < 0.01
    clk = bench.gpu_clocks(0)                                # no GPU here: None, never an exception
    assert clk is None or isinstance(clk, dict)
    assert bench.others_exit_status(bench.parse_args([])) == 0 and bench.others_exit_status(bench.parse_args(["--strict-exit"])) == 3


def test_selfcheck_formulas_and_fault_injection(monkeypatch):
    """cask_amd/selfcheck.py on CPU tensors with stand-in exchanges: a correct exchange passes 50 changing operands, a
    stale one is caught in its first odd exchange, and the injected fault (CASK_FAULT_STALE_HALO) makes a CORRECT
    exchange fail -- which is what proves the fallback path in the GPU dry runs."""
    import torch
    from cask_amd import selfcheck as sc
    n, world, rank = 96, 3, 1
    bounds = [0, 30, 70, 96]
    idx_all = torch.arange(n)
    assert not torch.equal(sc.operand(3, idx_all, n), sc.operand(4, idx_all, n))          # changes every exchange

    class FakePush:                                           # padded-stride all-gather done right, on the CPU
        S = 40

        def __init__(self, stale=False):
            self.stale, self.k, self.buf = stale, 0, [torch.zeros(world * self.S, dtype=torch.float64) for _ in range(2)]
            self.slot = torch.zeros(self.S, dtype=torch.float64)

        def own_slot(self):
            return self.slot

        def allgather(self, slot):
            e, out = self.k, self.buf[self.k % 2]
            self.k += 1
            for g in range(world):
                cnt = bounds[g + 1] - bounds[g]
                src_e = e - 1 if (self.stale and g != rank and e % 2 == 1) else e       # a peer whose slice is one exchange old
                val = slot[:cnt] if g == rank else sc.operand(src_e, torch.arange(bounds[g], bounds[g + 1]), n)
                out[g * self.S: g * self.S + cnt] = val
            return out

        def check(self):
            pass

        def allreduce(self, t):
            e = self.k
            self.k += 1
            total = torch.zeros_like(t)
            for g in range(world):
                total += t if g == rank else torch.tensor([self._contrib(g, e, j) for j in range(t.numel())], dtype=torch.float64)
            t.copy_(total)

        @staticmethod
        def _contrib(g, e, j):
            return (g + 1) * 0.125 + e * 1.0009765625 + j * 3.0 + ((5 * e + g) % 7) * 0.0625

    pos = torch.arange(world * FakePush.S)
    owner, off = pos // FakePush.S, pos % FakePush.S
    sizes = torch.tensor([bounds[g + 1] - bounds[g] for g in range(world)])
    valid = off < sizes[owner]
    idx_pad = torch.where(valid, torch.tensor(bounds[:-1])[owner] + off, torch.zeros_like(pos))
    idx_own = torch.arange(bounds[rank], bounds[rank + 1])
    ok, why = sc.check_push_allgather(torch, FakePush(), bounds[rank + 1] - bounds[rank], idx_own, idx_pad, valid, n)
    assert ok and why is None
    ok, why = sc.check_push_allgather(torch, FakePush(stale=True), bounds[rank + 1] - bounds[rank], idx_own, idx_pad, valid, n)
    assert not ok and "exchange 1" in why
    monkeypatch.setenv("CASK_FAULT_STALE_HALO", "push")
    ok, why = sc.check_push_allgather(torch, FakePush(), bounds[rank + 1] - bounds[rank], idx_own, idx_pad, valid, n)
    assert not ok and why.startswith("push all-gather:")
    monkeypatch.setenv("CASK_FAULT_STALE_HALO", "halo")                                  # another path's fault: not this one's
    assert sc.check_push_allgather(torch, FakePush(), bounds[rank + 1] - bounds[rank], idx_own, idx_pad, valid, n)[0]
    monkeypatch.delenv("CASK_FAULT_STALE_HALO")
    # all-reduce: rank-order sums (the fake adds in rank order only for rank 0's position; sums of these exact binary
    # fractions do not depend on the order)
    assert sc.check_push_allreduce(torch, FakePush(), rank, world, torch.device("cpu"))[0]
    monkeypatch.setenv("CASK_FAULT_STALE_HALO", "1")
    ok, why = sc.check_push_allreduce(torch, FakePush(), rank, world, torch.device("cpu"))
    assert not ok and "reduction 1" in why
    # the collective verdict: one failing rank sends everybody to the fallback, with its reason
    assert sc.agree(True, None, lambda v: v) == (True, None)
    assert sc.agree(True, None, lambda v: 0.0, lambda o: [None, "stale", "late"]) == (False, "rank 1: stale; rank 2: late")
