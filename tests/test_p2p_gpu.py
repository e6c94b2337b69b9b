"""Peer-to-peer exchange on the device: two (and three) processes, each with its own HIP context,
share x slices through cask_hip_shared_* and pull their halos with cask_hip_halo_pull_device.
The GPU box has one GPU, so the ranks share device 0 (control plane: gloo); the mapping, the
address table, the pull kernel and the extended-column product are exactly what runs with one GPU
per rank.  Results are compared with the oracle on the global matrix."""
import os
import socket

import numpy as np
import pytest

import oracle
from cask_amd import synth
from conftest import have_gpu, spawn_collect

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a GPU")]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cask_amd import dist as cdist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, rp, ci, va = case["matrix"]

        def fence():                               # host-side fence: gloo does not order device streams
            torch.cuda.synchronize()
            dist.barrier()

        sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, balance=case.get("balance", "nnz"),
                                           exchange="p2p", fence=fence, fused_halo=case.get("fused", False))
        assert sh.fused_halo == bool(case.get("fused", False))
        b0, b1 = sh.bounds[rank], sh.bounds[rank + 1]
        res = {"n_halo": sh.exchange.n_halo, "owners": sorted(sh.exchange.peers)}
        ys = []
        for x in case["xs"]:                       # several products: the halo is refreshed each time
            xl = torch.from_numpy(x[b0:b1].copy()).cuda()
            y = sh.spmv(xl)
            torch.cuda.synchronize()
            ys.append(y.cpu().numpy())
        res["ys"] = ys
        # in-place use: the solver's vector IS the shared slice, and the product is graph-captured
        sh.exchange.x_local.copy_(torch.from_numpy(case["xs"][0][b0:b1].copy()).cuda())
        y = torch.zeros(b1 - b0, dtype=torch.float64, device="cuda")
        fence()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            if not sh.fused_halo:
                sh.exchange.pull()
            sh.matrix.spmv_device(sh.exchange.x_ext, y)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            if not sh.fused_halo:
                sh.exchange.pull()
            sh.matrix.spmv_device(sh.exchange.x_ext, y)
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        res["y_graph"] = y.cpu().numpy()
        fence()
        if case.get("bicg") is not None:
            # BASELINE config 5 in small: BiCG with A and A^T row-sharded over the same shared vectors (6 slots),
            # both read their halos in-kernel; the passes run in the engine (cask_hip_solve_device)
            trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
            sh.close()
            sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, balance=case.get("balance", "nnz"),
                                               exchange="p2p", fence=fence, fused_halo=True, solver_slots=6)
            sht = cdist.ShardedSpmv.from_global(trp, tci, tva, n, rank, world, exchange="p2p", fence=fence,
                                                fused_halo=True, share_with=sh)
            bl = torch.from_numpy(case["bicg"][b0:b1].copy()).cuda()
            for mode in case.get("modes", (0,)):
                xs, it, conv = sh.bicg(sht, bl, tol=1e-9, mode=mode)
                torch.cuda.synchronize()
                res[f"bicg_x{mode}"], res[f"bicg_it{mode}"], res[f"bicg_conv{mode}"] = xs.cpu().numpy(), it, conv
                fence()
            sht.exchange = None                    # the vectors belong to sh
        sh.close()
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def run_world(world, case):
    return spawn_collect(_worker, (world, free_port(), case), world)


def check(world, matrix, balance="nnz", fused=False):
    from cask_amd import dist as cdist
    n, rp, ci, va = matrix
    rng = np.random.default_rng(7)
    xs = [np.arange(n, dtype=np.float64) * 0.25, rng.uniform(-1, 1, n), rng.standard_normal(n)]
    res = run_world(world, {"matrix": matrix, "xs": xs, "balance": balance, "fused": fused})
    bounds = cdist.partition_rows_by_nnz(rp, world) if balance == "nnz" else cdist.partition_rows_even(n, world)
    for k, x in enumerate(xs):
        exp = oracle.csr_spmv(rp, ci, va, x)
        got = np.concatenate([r["ys"][k] for r in res])
        oracle.assert_almost_equal(got, exp)
    got = np.concatenate([r["y_graph"] for r in res])
    oracle.assert_almost_equal(got, oracle.csr_spmv(rp, ci, va, xs[0]))
    return res, bounds


@pytest.mark.parametrize("fused", [False, True], ids=["pull", "in_kernel"])
def test_two_ranks_power_law_matrix(fused):
    """30 % of the columns are uniformly random: every rank pulls from every other rank."""
    res, _ = check(2, synth.webbase_like(n=20_000, nnz_target=70_000, max_row=900, seed=5), fused=fused)
    assert all(r["n_halo"] > 1000 for r in res)
    assert res[0]["owners"] == [1] and res[1]["owners"] == [0]


@pytest.mark.parametrize("fused", [False, True], ids=["pull", "in_kernel"])
def test_three_ranks_banded_matrix_even_split(fused):
    res, _ = check(3, synth.cant_like(n=9_000, per_row=17, band=200, seed=3), balance="even", fused=fused)
    assert res[1]["owners"] == [0, 2] and res[0]["owners"] == [1]      # a band couples neighbours only
    assert all(0 < r["n_halo"] <= 400 for r in res)


def test_two_ranks_block_diagonal_needs_no_halo():
    n, rp, ci, va = synth.cant_like(n=4_000, per_row=9, band=50, seed=1)
    # cut every coupling across the middle
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = (rows < n // 2) == (ci < n // 2)
    rp2 = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp2, rows[keep] + 1, 1)
    rp2 = np.cumsum(rp2).astype(np.int32)
    res, _ = check(2, (n, rp2, ci[keep], va[keep]), balance="even")
    assert all(r["n_halo"] == 0 and r["owners"] == [] for r in res)


def test_two_ranks_bicg_with_in_kernel_halos():
    """A and A^T products of a nonsymmetric stencil system sharded over two processes (balance by rows so
    that both operators use the same slices), dots all-reduced: same answer as the oracle's BiCG, in the composed
    form (the product launches compose p and pt; the two all-reduces of a pass are its only ordering) and in the
    classic form (separate p update + a fence collective)."""
    matrix = synth.small("atmosmodd", factor=32)
    n, rp, ci, va = matrix
    x0 = np.random.default_rng(3).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b, tol=1e-9)
    xs = [np.arange(n, dtype=np.float64) * 0.25]
    res = run_world(2, {"matrix": matrix, "xs": xs, "balance": "even", "fused": True, "bicg": b, "modes": (0, 1, 2)})
    for mode in (0, 1, 2):
        got = np.concatenate([r[f"bicg_x{mode}"] for r in res])
        assert want_conv and all(r[f"bicg_conv{mode}"] for r in res)
        assert all(abs(r[f"bicg_it{mode}"] - want_it) <= 1 for r in res), mode
        np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9)
