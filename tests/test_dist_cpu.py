"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row partitioning, the
all-gather of x (even and uneven slices), the all-reduced dots and the solver callbacks of
cask_amd/dist.py.  The local block product is injected (the CPU oracle -- tests may use it); on GPUs
it is the HIP engine."""
import os
import socket

import numpy as np
import pytest

import oracle
from cask_amd import dist as cdist
from cask_amd import synth


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, rp, ci, va = case["matrix"]
        bounds = case["bounds"]
        lrp, lci, lva = cdist.slice_rows(rp, ci, va, bounds[rank], bounds[rank + 1])

        sh = cdist.ShardedSpmv(bounds, rank, world, None, torch.device("cpu"))
        lci_pad = sh.pad_columns(lci)                            # columns address the padded gathered vector

        def local_product(x_full, y_local):
            assert x_full.numel() == sh.n_full == world * sh.S
            y_local.copy_(torch.from_numpy(oracle.csr_spmv(lrp, lci_pad, lva, x_full.numpy())))

        sh.local_product = local_product
        x = torch.from_numpy(case["x"][bounds[rank]:bounds[rank + 1]].copy())
        y = sh.spmv(x)
        res = {"y": y.numpy().copy(), "dot": float(sh.dot(x, x))}
        if case.get("cg"):
            # The engine's sharded CG (cask_hip_solve_device, classic form) restated on numpy in THIS test and
            # driven through the product code's real callbacks: the operand all-gather (uneven slices) and the
            # in-place all-reduce of scalars at a raw address.  Checks the collective protocol on CPU; the
            # arithmetic of the engine itself is tested on the GPU (tests/test_p2p_gpu.py).
            import ctypes
            allreduce, exchange = sh._allreduce_callback(), sh._exchange_callback()
            nl = bounds[rank + 1] - bounds[rank]
            b = case["b"][bounds[rank]:bounds[rank + 1]].copy()
            ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p).value          # noqa: E731
            full, scal = np.zeros(sh.n_full), np.zeros(4)
            slot = np.zeros(sh.S)                                # the engine's operand slot: stride doubles

            def product(v):
                slot[:nl] = v
                assert exchange(ptr(slot), ptr(full), 0) == 0    # ONE all_gather_into_tensor, no pad / copy steps
                np.testing.assert_array_equal(sh.unpad(torch.from_numpy(full)).numpy()[bounds[rank]:bounds[rank + 1]], v)
                return oracle.csr_spmv(lrp, lci_pad, lva, full)

            def allsum(v):
                scal[0] = v
                assert allreduce(ptr(scal), 1, 0) == 0
                return scal[0]

            xs = np.zeros(nl)
            r = b - product(xs)
            p = r.copy()
            rsold = allsum(r @ r)
            it, conv = 0, False
            for i in range(case.get("maxiters", 2000)):
                Ap = product(p)
                alpha = rsold / allsum(p @ Ap)
                xs += alpha * p
                r -= alpha * Ap
                rsnew = allsum(r @ r)
                if rsnew <= 1e-10:
                    conv = True
                    break
                p = r + (rsnew / rsold) * p
                rsold, it = rsnew, i
            res.update({"cg_x": xs, "cg_it": it, "cg_conv": conv})
        out[rank] = res
    finally:
        dist.destroy_process_group()


def run_world(world, case):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, free_port(), case, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def test_partition_by_nnz_balances_work():
    n, rp, ci, va = synth.small("webbase-1M", factor=32)
    for world in (2, 3, 8):
        b = cdist.partition_rows_by_nnz(rp, world)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
        work = [(rp[b[g + 1]] - rp[b[g]]) + (b[g + 1] - b[g]) for g in range(world)]
        assert max(work) <= 1.05 * (sum(work) / world) + synth.row_stats(rp)["row_max"] + 1
    assert cdist.partition_rows_even(10, 3) == [0, 3, 6, 10]          # remainder to the last block (Spmv.cpp:353-364)
    b0 = cdist.partition_rows_by_nnz(np.zeros(6, dtype=np.int32), 2)             # all-empty rows still split
    assert b0[0] == 0 and b0[-1] == 5 and 0 < b0[1] < 5


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_spmv_uneven_slices(world):
    n, rp, ci, va = synth.small("webbase-1M", factor=64)
    x = np.random.default_rng(0).uniform(-1, 1, n)
    bounds = cdist.partition_rows_by_nnz(rp, world)
    assert len({bounds[g + 1] - bounds[g] for g in range(world)}) > 1          # genuinely uneven
    res = run_world(world, {"matrix": (n, rp, ci, va), "bounds": bounds, "x": x})
    want = oracle.csr_spmv(rp, ci, va, x)
    got = np.concatenate([r["y"] for r in res])
    assert np.array_equal(got, want)                                           # same sequential order per row
    for r in res:
        assert abs(r["dot"] - float(x @ x)) <= 1e-12 * float(x @ x)


def test_sharded_spmv_even_slices_weak_scaling_blocks():
    """The bench's weak-scaling matrix: one cant-like block per rank, seams coupled."""
    world, nl = 2, 1500
    blocks = [synth.cant_like_shard(g, world, n_local=nl, n_couple=200) for g in range(world)]
    n = world * nl
    rp = np.concatenate([[0]] + [b[2][1:] + sum(bb[2][-1] for bb in blocks[:g]) for g, b in enumerate(blocks)]).astype(np.int32)
    ci = np.concatenate([b[3] for b in blocks])
    va = np.concatenate([b[4] for b in blocks])
    assert ci.max() < n and (ci[: blocks[0][2][-1]] >= nl).any()               # block 0 references block 1's columns
    x = np.arange(n) * 0.25 / n
    res = run_world(world, {"matrix": (n, rp, ci, va), "bounds": [0, nl, n], "x": x})
    assert np.array_equal(np.concatenate([r["y"] for r in res]), oracle.csr_spmv(rp, ci, va, x))


def test_solver_callbacks_carry_a_distributed_cg():
    """The all-reduce and operand-exchange callbacks the sharded solvers hand to cask_hip_solve_device,
    exercised by a numpy restatement of its classic pass (the engine itself needs a GPU)."""
    n, rp, ci, va = synth.small("cant", factor=64)
    x0 = np.arange(n) * 0.25 / n
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b)
    bounds = cdist.partition_rows_by_nnz(rp, 2)
    res = run_world(2, {"matrix": (n, rp, ci, va), "bounds": bounds, "x": x0, "b": b, "cg": True})
    got = np.concatenate([r["cg_x"] for r in res])
    assert all(r["cg_conv"] == want_conv for r in res)
    assert all(abs(r["cg_it"] - want_it) <= 1 for r in res), ([r["cg_it"] for r in res], want_it)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9)


def test_world_eight_sharded_spmv_and_distributed_cg():
    """World 8 -- BASELINE configs[3] / [4] are "across 8 x MI355X" and a session may put at most 6 processes on its one
    GPU (profiles/r06_world_dryrun.txt), so the 8-rank control flow runs here, over gloo: nnz-balanced (uneven) row
    blocks of the power-law look-alike, ONE padded-stride all-gather per product, the all-reduced dots, and the sharded
    CG driven through the solver's real callbacks to the oracle's iteration count."""
    world = 8
    n, rp, ci, va = synth.small("webbase-1M", factor=32)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    bounds = cdist.partition_rows_by_nnz(rp, world)
    assert len(bounds) == world + 1 and len({bounds[g + 1] - bounds[g] for g in range(world)}) > 1
    res = run_world(world, {"matrix": (n, rp, ci, va), "bounds": bounds, "x": x})
    assert len(res) == world
    assert np.array_equal(np.concatenate([r["y"] for r in res]), oracle.csr_spmv(rp, ci, va, x))
    for r in res:
        assert abs(r["dot"] - float(x @ x)) <= 1e-12 * float(x @ x)
    n, rp, ci, va = synth.small("cant", factor=32)
    x0 = np.arange(n) * 0.25 / n
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b)
    bounds = cdist.partition_rows_by_nnz(rp, world)
    res = run_world(world, {"matrix": (n, rp, ci, va), "bounds": bounds, "x": x0, "b": b, "cg": True})
    assert all(r["cg_conv"] == want_conv for r in res)
    assert all(abs(r["cg_it"] - want_it) <= 1 for r in res), ([r["cg_it"] for r in res], want_it)
    np.testing.assert_allclose(np.concatenate([r["cg_x"] for r in res]), want, rtol=1e-7, atol=1e-9)


def test_transpose_csr_matches_oracle():
    n, rp, ci, va = synth.small("atmosmodd", factor=64)
    trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
    x = np.random.default_rng(1).standard_normal(n)
    assert np.array_equal(oracle.csr_spmv(trp, tci, tva, x), oracle.csr_spmv_t(n, rp, ci, va, x))
    assert all(np.all(np.diff(tci[trp[r]:trp[r + 1]]) > 0) for r in range(0, n, max(1, n // 50)))


def _selfcheck_worker(rank, world, port, out):
    """check_fused_halo over gloo with stand-in products: every rank's "fused product" reads the OTHER rank's shared
    slice (a file-free stand-in: the slices are all-gathered in the fence), so a fault injected on ONE rank fails its
    READER only."""
    import torch
    import torch.distributed as dist
    from cask_amd import selfcheck as sc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nl = 16
        n = world * nl
        idx_own = torch.arange(rank * nl, (rank + 1) * nl)
        mine = torch.zeros(nl, dtype=torch.float64)
        seen = [torch.zeros(nl, dtype=torch.float64) for _ in range(world)]
        fences = [0]

        def fence():                                          # a collective: barrier + what the peers' slices hold now
            fences[0] += 1
            dist.all_gather(seen, mine)

        def plain_ref(e):                                     # y = sum of the peers' operands, from the formula
            return sum(sc.operand(e, torch.arange(g * nl, (g + 1) * nl), n) for g in range(world) if g != rank)

        def fused(y):                                         # ... from what the peers' slices really hold
            y.copy_(sum(seen[g] for g in range(world) if g != rank))

        ok, why = sc.check_fused_halo(torch, fused, plain_ref, mine, idx_own, fence, n, n=12, rank=rank)
        t = torch.tensor([1.0 if ok else 0.0])
        agreed, reason = sc.agree(ok, why, lambda v: (dist.all_reduce(t.fill_(v), op=dist.ReduceOp.MIN), float(t[0]))[1],
                                  lambda o: (lambda l: (dist.all_gather_object(l, o), l)[1])([None] * world))
        out[rank] = {"ok": ok, "why": why, "fences": fences[0], "agreed": agreed, "reason": reason}
    finally:
        dist.destroy_process_group()


def test_selfcheck_with_a_fault_on_one_rank_only_completes_its_collectives(monkeypatch):
    """a halo fault is seen by the readers of ONE slice; the rank that sees it must finish all
    2 n fences like its peers before the verdict's collectives, or they would be mismatched (a hang where the fallback
    should engage).  Fault on rank 1 only: rank 0 (its reader) fails, rank 1 passes, both ran 2 n fences, both agree."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("CASK_FAULT_STALE_HALO", "halo")
    monkeypatch.setenv("CASK_FAULT_RANK", "1")
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_selfcheck_worker, args=(2, free_port(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert not r0["ok"] and "exchange 1" in r0["why"]         # the first odd exchange, and the loop went on
    assert r1["ok"] and r1["why"] is None
    assert r0["fences"] == r1["fences"] == 24
    assert not r0["agreed"] and not r1["agreed"] and r0["reason"] == r1["reason"] and r0["reason"].startswith("rank 0:")
    monkeypatch.delenv("CASK_FAULT_STALE_HALO")
    out2 = mgr.dict()
    mp.spawn(_selfcheck_worker, args=(2, free_port(), out2), nprocs=2, join=True)
    assert out2[0]["agreed"] and out2[1]["agreed"] and out2[0]["fences"] == 24
