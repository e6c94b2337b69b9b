"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row partitioning, the
all-gather of x (even and uneven slices), the all-reduced dots and the distributed CG of
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

        def local_product(x_full, y_local):
            y_local.copy_(torch.from_numpy(oracle.csr_spmv(lrp, lci, lva, x_full.numpy())))

        sh = cdist.ShardedSpmv(bounds, rank, world, local_product, torch.device("cpu"))
        x = torch.from_numpy(case["x"][bounds[rank]:bounds[rank + 1]].copy())
        y = sh.spmv(x)
        res = {"y": y.numpy().copy(), "dot": float(sh.dot(x, x))}
        if case.get("cg"):
            b = torch.from_numpy(case["b"][bounds[rank]:bounds[rank + 1]].copy())
            xs, it, conv = sh.cg(b, maxiters=case.get("maxiters", 2000))
            res.update({"cg_x": xs.numpy().copy(), "cg_it": it, "cg_conv": conv})
        if case.get("bicg"):
            trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
            tl = cdist.slice_rows(trp, tci, tva, bounds[rank], bounds[rank + 1])

            def local_product_t(x_full, y_local):
                y_local.copy_(torch.from_numpy(oracle.csr_spmv(tl[0], tl[1], tl[2], x_full.numpy())))

            sht = cdist.ShardedSpmv(bounds, rank, world, local_product_t, torch.device("cpu"))
            b = torch.from_numpy(case["b"][bounds[rank]:bounds[rank + 1]].copy())
            xs, it, conv = sh.bicg(sht, b, tol=case.get("tol", 1e-5))
            res.update({"bicg_x": xs.numpy().copy(), "bicg_it": it, "bicg_conv": conv})
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


def test_distributed_cg_matches_oracle():
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


def test_transpose_csr_matches_oracle():
    n, rp, ci, va = synth.small("atmosmodd", factor=64)
    trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
    x = np.random.default_rng(1).standard_normal(n)
    assert np.array_equal(oracle.csr_spmv(trp, tci, tva, x), oracle.csr_spmv_t(n, rp, ci, va, x))
    assert all(np.all(np.diff(tci[trp[r]:trp[r + 1]]) > 0) for r in range(0, n, max(1, n // 50)))


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_bicg_matches_oracle(world):
    """BASELINE config 5 in small: nonsymmetric atmosmodd-like system, A and A^T products row-sharded."""
    n, rp, ci, va = synth.small("atmosmodd", factor=64)
    x0 = np.random.default_rng(2).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b, tol=1e-9)
    bounds = cdist.partition_rows_by_nnz(rp, world)
    res = run_world(world, {"matrix": (n, rp, ci, va), "bounds": bounds, "x": x0, "b": b, "bicg": True, "tol": 1e-9})
    got = np.concatenate([r["bicg_x"] for r in res])
    assert want_conv and all(r["bicg_conv"] for r in res)
    assert all(abs(r["bicg_it"] - want_it) <= 1 for r in res), ([r["bicg_it"] for r in res], want_it)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(got, x0, rtol=1e-6, atol=1e-8)

