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
from conftest import have_gpu, spawn_collect

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


def _worker(rank, world, port, cases, out):
    """One launch, a LIST of cases: the ranks start (torch import, HIP context, gloo rendezvous: 2-3 s) once per world size
    and run the module's cases of that size one after the other, each on operators of its own."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        results = []
        for case in cases:
            # dot products reduced by peer stores (cask_hip_push_allreduce): opt-in, read when an operator first solves
            os.environ.pop("CASK_PEER_ALLREDUCE", None)
            if case.get("peer_allreduce"):
                os.environ["CASK_PEER_ALLREDUCE"] = "1"
            results.append(_run_case(rank, world, case))
        out.put((rank, results))
    finally:
        dist.destroy_process_group()


def _run_case(rank, world, case):
    import torch
    import torch.distributed as dist
    from cask_amd import dist as cdist
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
    return res


def run_world_batch(world, cases):
    """-> [case][rank] results."""
    by_rank = spawn_collect(_worker, (world, free_port(), list(cases)), world)
    return [[by_rank[r][c] for r in range(world)] for c in range(len(cases))]


def run_world(world, case):
    return run_world_batch(world, [case])[0]


# ---- the module's cases by world size: ONE launch per size, results handed to the tests below -----------------------
G3, AT = ("small", "G3_circuit", 64), ("small", "atmosmodd", 64)


def _initial_guess_inputs():
    """(b, x_init) for CG on the G3-like system and for BiCG on the stencil (one generator, in this order)."""
    rng = np.random.default_rng(17)
    n_g3, n_at = _matrix(G3)[0], _matrix(AT)[0]
    cg = (rng.standard_normal(n_g3), rng.uniform(-1, 1, n_g3))
    bicg = (rng.standard_normal(n_at), rng.uniform(-1, 1, n_at))
    return cg, bicg


def _not_converged_rhs():
    return np.random.default_rng(7).standard_normal(_matrix(G3)[0])


def _chain_x0():
    return np.random.default_rng(11).uniform(-1, 1, _matrix(AT)[0])


def _cases(world):
    (b_cg, x_cg), (b_bi, x_bi) = _initial_guess_inputs()
    cases = {}
    if world in (2, 3):
        cases["cg_p2p"] = {"matrix": G3, "exchange": "p2p", "b": "A*x0", "modes": (0, 1, 2)}
        cases["cg_all_gather"] = {"matrix": G3, "exchange": "all_gather", "b": "A*x0", "modes": (0,)}
    if world == 2:
        for ex in ("p2p", "all_gather"):
            cases[f"not_converged_{ex}"] = {"matrix": G3, "exchange": ex, "b": _not_converged_rhs(), "maxiters": 5}
            cases[f"guess_bicg_{ex}"] = {"matrix": AT, "exchange": ex, "solver": "bicg", "b": b_bi, "x_init": x_bi, "tol": 1e-9}
    if world == 3:
        for ex in ("p2p", "all_gather"):
            cases[f"bicg_{ex}"] = {"matrix": AT, "exchange": ex, "solver": "bicg", "b": "A*x0", "tol": 1e-9}
            cases[f"guess_cg_{ex}"] = {"matrix": G3, "exchange": ex, "b": b_cg, "x_init": x_cg,
                                       "modes": (1, 2) if ex == "p2p" else (0,)}
        cases["peer_cg"] = {"matrix": G3, "exchange": "p2p", "b": "A*x0", "peer_allreduce": True, "modes": (0,)}
        cases["peer_bicg"] = {"matrix": AT, "exchange": "p2p", "b": "A*x0", "peer_allreduce": True, "solver": "bicg", "tol": 1e-9}
        cases["chain"] = {"matrix": AT, "exchange": "p2p", "chain": 6, "chain_x0": _chain_x0()}
    if world == 5:
        for name, ex in (("webbase-1M", "all_gather"), ("atmosmodd", "p2p")):
            n = synth.SPECS[name][0]
            cases[name] = {"matrix": ("full", name), "exchange": ex, "spmv_x": np.arange(n, dtype=np.float64) * 0.25 / n}
    return cases


_RESULTS = {}


def world_results(world, key):
    if world not in _RESULTS:
        cases = _cases(world)
        _RESULTS[world] = dict(zip(cases, run_world_batch(world, list(cases.values()))))
    return _RESULTS[world][key]


@pytest.mark.parametrize("world,exchange", [(2, "p2p"), (3, "p2p"), (2, "all_gather"), (3, "all_gather")])
def test_sharded_cg_matches_oracle(world, exchange):
    """Config 3 in small, row-sharded: composed passes over in-kernel halos (p2p; also forced classic, which spends a
    third collective per pass as a fence) and classic passes with the operand all-gathered (uneven slices)."""
    n, rp, ci, va = _matrix(G3)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b)
    modes = (0, 1, 2) if exchange == "p2p" else (0,)
    res = world_results(world, f"cg_{exchange}")
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
    n, rp, ci, va = _matrix(AT)
    x0 = np.random.default_rng(5).uniform(-1, 1, n)
    b = oracle.csr_spmv(rp, ci, va, x0)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b, tol=1e-9)
    res = world_results(3, f"bicg_{exchange}")
    got = np.concatenate([r["x0"] for r in res])
    assert want_conv and all(r["conv0"] for r in res)
    assert all(abs(r["it0"] - want_it) <= 1 for r in res), ([r["it0"] for r in res], want_it)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(got, x0, rtol=1e-6, atol=1e-8)


def test_sharded_cg_not_converged_reports_last_iteration():
    n, rp, ci, va = _matrix(G3)
    b = _not_converged_rhs()
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b, maxiters=5)
    for exchange in ("p2p", "all_gather"):
        res = world_results(2, f"not_converged_{exchange}")
        assert not want_conv and not any(r["conv0"] for r in res)
        assert all(r["it0"] == want_it == 4 for r in res)        # iterations = i of the last pass (:231)
        np.testing.assert_allclose(np.concatenate([r["x0"] for r in res]), want, rtol=1e-9, atol=1e-12)


def test_operand_that_changes_every_product_in_kernel_halo():
    """x_{k+1} = A x_k / 4 + x_k over 6 products on three ranks, halos read inside the product kernel from slices
    the owners rewrite between products: any halo entry served one product late shows in the final iterate."""
    n, rp, ci, va = _matrix(AT)
    x = _chain_x0()
    res = world_results(3, "chain")
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
    res = world_results(5, name)
    assert all(r["exchange"] == ("p2p_fused" if exchange == "p2p" else "all_gather") for r in res)
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
        out.put((rank, res))
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
    res = spawn_collect(_tiny_worker, (3, free_port()), 3)
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
    (b, x_init), (b_bi, x_bi) = _initial_guess_inputs()
    n, rp, ci, va = _matrix(G3)
    want, want_it, want_conv = oracle.cg_full(rp, ci, va, b, x0=x_init)
    modes = (1, 2) if exchange == "p2p" else (0,)
    res = world_results(3, f"guess_cg_{exchange}")
    for mode in modes:
        got = np.concatenate([r[f"x{mode}"] for r in res])
        assert all(r[f"conv{mode}"] == want_conv for r in res)
        assert all(abs(r[f"it{mode}"] - want_it) <= 2 for r in res), ([r[f"it{mode}"] for r in res], want_it)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
    # BiCG, nonsymmetric, same thing
    n, rp, ci, va = _matrix(AT)
    want, want_it, want_conv = oracle.bicg(rp, ci, va, b_bi, x0=x_bi, tol=1e-9)
    res = world_results(2, f"guess_bicg_{exchange}")
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
    res = world_results(3, f"peer_{solver}")
    got = np.concatenate([r["x0"] for r in res])
    assert want_conv and all(r["conv0"] for r in res)
    assert all(abs(r["it0"] - want_it) <= 2 for r in res) and len({r["it0"] for r in res}) == 1
    assert all(r.get("collectives", "").startswith("peer-store") for r in res), [r.get("collectives") for r in res]
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(want).max()))
