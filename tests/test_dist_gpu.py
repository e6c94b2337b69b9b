"""Row-sharded CG / BiCG and products on the device: several processes, each with its own HIP context, run
`cask_hip_solve_device` on their row block -- the engine's own kernels and recurrences, dot products all-reduced
through the callback.  The GPU box has ONE GPU, so the ranks share device 0 and the control plane is gloo (RCCL
refuses two ranks on one device; its path is exercised at world size 1 in tests/test_bench_gpu.py); the kernels,
the shared vector slots, the address tables and the order of collectives are exactly what runs with one GPU per
rank.  The pool allows at most 6 processes on the card, which bounds the world size here.  Every result is compared
with the oracle on the GLOBAL matrix."""
import os
import socket

import numpy as np
import pytest

import oracle
from cask_amd import synth
from conftest import have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a GPU")]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _matrix(spec):
    kind = spec[0]
    if kind == "small":
        return synth.small(spec[1], factor=spec[2])
    if kind == "full":
        return synth.GENERATORS[spec[1]]()
    raise KeyError(kind)


def _worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cask_amd import capi
    from cask_amd import dist as cdist
    torch.cuda.set_device(0)
    if case.get("peer_allreduce"):                 # dot products reduced by peer stores (cask_hip_push_allreduce), opt-in
        os.environ["CASK_PEER_ALLREDUCE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, rp, ci, va = _matrix(case["matrix"])

        def fence():                               # host-side fence: gloo does not order device streams
            torch.cuda.synchronize()
            dist.barrier()

        res = {}
        bicg = case.get("solver") == "bicg"
        kw = dict(balance=case.get("balance", "nnz"))
        if case["exchange"] == "p2p":
            kw.update(exchange="p2p", fence=fence, fused_halo=True, solver_slots=6 if bicg else 3)
        sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, **kw)
        b0, b1 = sh.bounds[rank], sh.bounds[rank + 1]
        res["bounds"] = (b0, b1)
        sht = None
        if bicg:
            trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
            if case["exchange"] == "p2p":
                sht = cdist.ShardedSpmv.from_global(trp, tci, tva, n, rank, world, exchange="p2p", fence=fence,
                                                    fused_halo=True, share_with=sh)
            else:
                sht = cdist.ShardedSpmv.from_global(trp, tci, tva, n, rank, world, bounds=sh.bounds)
        if case.get("b") is not None:
            b = np.asarray(case["b"]) if not isinstance(case["b"], str) else None
            if b is None:                          # "A*x0": the right-hand side of the reference's harness
                x0 = np.random.default_rng(5).uniform(-1, 1, n)
                b = oracle.csr_spmv(rp, ci, va, x0)
            bl = torch.from_numpy(b[b0:b1].copy()).cuda()
            x_init = None
            if case.get("x_init") is not None:                 # a non-zero initial guess: the set-up product reads it
                x_init = torch.from_numpy(np.asarray(case["x_init"])[b0:b1].copy()).cuda()   # through the halo too
            for mode in case.get("modes", (0,)):
                if bicg:
                    xs, it, conv = sh.bicg(sht, bl, x_init, tol=case.get("tol", 1e-5), mode=mode, maxiters=case.get("maxiters", 2000))
                else:
                    xs, it, conv = sh.cg(bl, x_init, tol=case.get("tol", 1e-5), mode=mode, maxiters=case.get("maxiters", 2000))
                torch.cuda.synchronize()
                res[f"x{mode}"], res[f"it{mode}"], res[f"conv{mode}"] = xs.cpu().numpy(), it, conv
                res[f"us{mode}"] = sh.last_usec_per_iteration
                res["collectives"] = sh.last_collectives
                fence()
        if case.get("chain"):
            # a product whose operand changes every time: x_{k+1} = y_k / 4 + x_k, every rank rewriting its shared
            # slice between products (a stale halo entry -- one product late -- changes every later iterate)
            x = torch.from_numpy(case["chain_x0"][b0:b1].copy()).cuda()
            for _ in range(case["chain"]):
                y = sh.spmv(x)
                x = y * 0.25 + x
            torch.cuda.synchronize()
            res["chain"] = x.cpu().numpy()
            fence()
        if case.get("spmv_x") is not None:
            xl = torch.from_numpy(np.asarray(case["spmv_x"])[b0:b1].copy()).cuda()
            y = sh.spmv(xl)
            torch.cuda.synchronize()
            # checked here, against this rank's rows of the oracle product (the full-size cases)
            want = oracle.csr_spmv(*cdist.slice_rows(rp, ci, va, b0, b1), np.asarray(case["spmv_x"]))
            bad, first = oracle.mismatches(y.cpu().numpy(), want)
            res["rows_wrong"] = bad
            res["exchange"] = "p2p_fused" if sh.exchange is not None else "all_gather"
            fence()
        if sht is not None and case["exchange"] == "p2p":
            sht.exchange = None                    # the vectors belong to sh
        sh.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def run_world(world, case):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, free_port(), case, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


@pytest.mark.parametrize("world,exchange", [(2, "p2p"), (3, "p2p"), (2, "all_gather"), (3, "all_gather")])
def test_sharded_cg_matches_oracle(world, exchange):
    """Config 3 in small, row-sharded: composed passes over in-kernel halos (p2p; also forced classic, which spends a
    third collective per pass as a fence) and classic passes with the operand all-gathered (uneven slices)."""
    spec = ("small", "G3_circuit", 64)
    n, rp, ci, va = _matrix(spec)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b)
    modes = (0, 1, 2) if exchange == "p2p" else (0,)
    res = run_world(world, {"matrix": spec, "exchange": exchange, "b": "A*x0", "modes": modes})
    assert want_conv
    for mode in modes:
        got = np.concatenate([r[f"x{mode}"] for r in res])
        assert all(r[f"conv{mode}"] for r in res)
        assert all(abs(r[f"it{mode}"] - want_it) <= 2 for r in res), ([r[f"it{mode}"] for r in res], want_it)
        assert len({r[f"it{mode}"] for r in res}) == 1          # every rank stops in the same pass
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
        assert np.linalg.norm(b - oracle.csr_spmv(rp, ci, va, got)) <= 2e-5


@pytest.mark.parametrize("exchange", ["p2p", "all_gather"])
def test_sharded_bicg_matches_oracle_three_ranks(exchange):
    """Config 5 in small on three ranks (nnz-balanced, uneven slices): A and A^T blocks over the same vectors."""
    spec = ("small", "atmosmodd", 64)
    n, rp, ci, va = _matrix(spec)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b, tol=1e-9)
    res = run_world(3, {"matrix": spec, "exchange": exchange, "solver": "bicg", "b": "A*x0", "tol": 1e-9})
    got = np.concatenate([r["x0"] for r in res])
    assert want_conv and all(r["conv0"] for r in res)
    assert all(abs(r["it0"] - want_it) <= 1 for r in res), ([r["it0"] for r in res], want_it)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(got, x0, rtol=1e-6, atol=1e-8)


def test_sharded_cg_not_converged_reports_last_iteration():
    spec = ("small", "G3_circuit", 64)
    n, rp, ci, va = _matrix(spec)
    b = np.random.default_rng(7).standard_normal(n)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b, maxiters=5)
    for exchange in ("p2p", "all_gather"):
        res = run_world(2, {"matrix": spec, "exchange": exchange, "b": b, "maxiters": 5})
        assert not want_conv and not any(r["conv0"] for r in res)
        assert all(r["it0"] == want_it == 4 for r in res)        # iterations = i of the last pass (:231)
        np.testing.assert_allclose(np.concatenate([r["x0"] for r in res]), want, rtol=1e-9, atol=1e-12)


def test_operand_that_changes_every_product_in_kernel_halo():
    """x_{k+1} = A x_k / 4 + x_k over 6 products on three ranks, halos read inside the product kernel from slices
    the owners rewrite between products: any halo entry served one product late shows in the final iterate."""
    spec = ("small", "atmosmodd", 64)
    n, rp, ci, va = _matrix(spec)
    x = np.random.default_rng(11).uniform(-1, 1, n)
    res = run_world(3, {"matrix": spec, "exchange": "p2p", "chain": 6, "chain_x0": x})
    want = x.copy()
    for _ in range(6):
        want = oracle.csr_spmv(rp, ci, va, want) * 0.25 + want
    got = np.concatenate([r["chain"] for r in res])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max())


def test_one_rank_sharded_solver_is_the_single_gpu_solver():
    """world = 1: no callbacks, the same code path and the same bits as cask_hip_cg."""
    import torch
    from cask_amd import capi
    from cask_amd import dist as cdist
    torch.cuda.set_device(0)
    n, rp, ci, va = synth.small("G3_circuit", factor=16)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    want, want_it, want_conv, _ = m.cg(b)
    m.close()
    sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, 0, 1)
    got, it, conv = sh.cg(torch.from_numpy(b).cuda())
    assert (it, conv) == (want_it, want_conv)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("name,exchange", [("webbase-1M", "all_gather"), ("atmosmodd", "p2p")])
def test_full_size_configs_in_five_blocks(name, exchange):
    """BASELINE configs 4 and 5 at FULL size, row-partitioned by nnz over 5 processes sharing the GPU (the pool's
    process limit; the 8-way split of the same matrices is checked block by block in test_eight_way_partition):
    the exchange each workload gets from bench.py -- RCCL-style all-gather of x for the power-law matrix whose
    halo is nearly all of x, the in-kernel halo for the stencil -- every rank's rows against the oracle."""
    n = synth.SPECS[name][0]
    x = np.arange(n, dtype=np.float64) * 0.25 / n
    res = run_world(5, {"matrix": ("full", name), "exchange": exchange, "spmv_x": x})
    assert sum(r["rows_wrong"] for r in res) == 0
    assert res[0]["bounds"][0] == 0 and res[-1]["bounds"][1] == n
    assert all(res[g]["bounds"][1] == res[g + 1]["bounds"][0] for g in range(4))


@pytest.mark.parametrize("name", ["webbase-1M", "atmosmodd"])
def test_eight_way_partition_block_by_block(name):
    """The 8 row blocks bench.py deals to 8 GPUs, one after the other on this one: block g of the nnz-balanced
    partition with global columns, y_g = A_g x, against the oracle's rows."""
    import torch
    from cask_amd import capi
    from cask_amd import dist as cdist
    torch.cuda.set_device(0)
    n, rp, ci, va = synth.GENERATORS[name]()
    x = np.random.default_rng(8).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    bounds = cdist.partition_rows_by_nnz(rp, 8)
    work = [(rp[bounds[g + 1]] - rp[bounds[g]]) + (bounds[g + 1] - bounds[g]) for g in range(8)]
    assert max(work) <= 1.02 * sum(work) / 8 + synth.row_stats(rp)["row_max"]
    xt = torch.from_numpy(x).cuda()
    for g in range(8):
        lrp, lci, lva = cdist.slice_rows(rp, ci, va, bounds[g], bounds[g + 1])
        m = capi.CsrMatrix.from_host(bounds[g + 1] - bounds[g], n, lrp, lci, lva)
        y = torch.empty(bounds[g + 1] - bounds[g], dtype=torch.float64, device="cuda")
        m.spmv_device(xt, y)
        torch.cuda.synchronize()
        m.close()
        oracle.assert_almost_equal(y.cpu().numpy(), want[bounds[g]:bounds[g + 1]], what=f"{name} block {g}/8")


def _tiny_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cask_amd import capi
    from cask_amd import dist as cdist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, rp, ci, va, b = _tiny_system()
        bounds = [0, 3, 6, 7]                       # rank 2 owns ONE row with ONE nonzero: no MERGE plan there

        def fence():
            torch.cuda.synchronize()
            dist.barrier()

        res = {}
        try:
            cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, exchange="p2p", fence=fence, fused_halo=True,
                                          solver_slots=3, bounds=bounds)
            res["p2p"] = "built"
        except capi.CaskHipError as e:
            res["p2p"] = "refused: " + str(e)[:40]
        sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, bounds=bounds)
        res["fuses_dot"] = bool(sh.matrix.info.fuses_dot)
        bl = torch.from_numpy(b[bounds[rank]:bounds[rank + 1]].copy()).cuda()
        xs, it, conv = sh.cg(bl, tol=1e-12)
        torch.cuda.synchronize()
        res.update(x=xs.cpu().numpy(), it=it, conv=conv)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def _tiny_system():
    import scipy.sparse as sp
    a = sp.diags([np.arange(2.0, 9.0)], [0], format="lil")
    for i, j, v in ((0, 1, 0.5), (1, 2, -0.25), (2, 4, 0.125), (3, 5, 0.5), (0, 5, -0.5)):
        a[i, j] = v
        a[j, i] = v
    a = sp.csr_matrix(a)
    a.sort_indices()
    x0 = np.arange(1.0, 8.0)
    return 7, a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64), a @ x0


def test_a_rank_without_a_merge_plan_does_not_strand_the_others():
    """Every rank of a sharded solve must take the same form of pass.  A block with fewer than 2 nonzeros runs the
    VECTOR kernel (no dot epilogue, no in-kernel halo): the in-kernel-halo construction is then refused on EVERY rank
    (a refusal on one rank alone would leave the others in a collective), and the all-gather solver agrees on the
    classic pass collectively."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tiny_worker, args=(3, free_port(), out), nprocs=3, join=True)
    res = [out[r] for r in range(3)]
    assert all(r["p2p"].startswith("refused") for r in res), [r["p2p"] for r in res]
    assert [r["fuses_dot"] for r in res] == [True, True, False]
    n, rp, ci, va, b = _tiny_system()
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b, tol=1e-12)
    assert want_conv and all(r["conv"] for r in res) and len({r["it"] for r in res}) == 1
    assert abs(res[0]["it"] - want_it) <= 1
    np.testing.assert_allclose(np.concatenate([r["x"] for r in res]), np.arange(1.0, 8.0), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("exchange", ["p2p", "all_gather"])
def test_sharded_solvers_with_a_nonzero_initial_guess(exchange):
    """r = b - A x0 with x0 != 0: the set-up product reads the initial guess of every rank (through the halo table
    with the slot offset, or the all-gather) before the first pass."""
    spec = ("small", "G3_circuit", 64)
    n, rp, ci, va = _matrix(spec)
    rng = np.random.default_rng(17)
    b, x_init = rng.standard_normal(n), rng.uniform(-1, 1, n)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b, x0=x_init)
    modes = (1, 2) if exchange == "p2p" else (0,)
    res = run_world(3, {"matrix": spec, "exchange": exchange, "b": b, "x_init": x_init, "modes": modes})
    for mode in modes:
        got = np.concatenate([r[f"x{mode}"] for r in res])
        assert all(r[f"conv{mode}"] == want_conv for r in res)
        assert all(abs(r[f"it{mode}"] - want_it) <= 2 for r in res), ([r[f"it{mode}"] for r in res], want_it)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
    # BiCG, nonsymmetric, same thing
    spec = ("small", "atmosmodd", 64)
    n, rp, ci, va = _matrix(spec)
    b, x_init = rng.standard_normal(n), rng.uniform(-1, 1, n)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b, x0=x_init, tol=1e-9)
    res = run_world(2, {"matrix": spec, "exchange": exchange, "solver": "bicg", "b": b, "x_init": x_init, "tol": 1e-9})
    got = np.concatenate([r["x0"] for r in res])
    assert want_conv and all(r["conv0"] for r in res) and all(abs(r["it0"] - want_it) <= 1 for r in res)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("solver", ["cg", "bicg"])
def test_sharded_solvers_with_peer_store_allreduce(solver):
    """CASK_PEER_ALLREDUCE=1: the dot products of a row-sharded pass are summed locally and reduced across the ranks by
    ONE launch (k_push_sum_allreduce: 16-byte {value, sequence} granules stored into every peer's table, rank-order sum)
    -- no collective library in the pass.  Three ranks, composed passes over in-kernel halos; iteration counts and the
    solution against the oracle, every rank stopping in the same pass."""
    spec = ("small", "G3_circuit", 64) if solver == "cg" else ("small", "atmosmodd", 64)
    n, rp, ci, va = _matrix(spec)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    tol = 1e-5 if solver == "cg" else 1e-9
    want, want_it, want_conv = (oracle.cg_full(rp, ci, va, b) if solver == "cg" else oracle.bicg(rp, ci, va, b, tol=tol))
    case = {"matrix": spec, "exchange": "p2p", "b": "A*x0", "peer_allreduce": True}
    if solver == "bicg":
        case.update(solver="bicg", tol=tol)
    else:
        case["modes"] = (0,)
    res = run_world(3, case)
    got = np.concatenate([r["x0"] for r in res])
    assert want_conv and all(r["conv0"] for r in res)
    assert all(abs(r["it0"] - want_it) <= 2 for r in res) and len({r["it0"] for r in res}) == 1
    assert all(r.get("collectives", "").startswith("peer-store") for r in res), [r.get("collectives") for r in res]
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(want).max()))
