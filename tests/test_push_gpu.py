"""The push all-gather (include/cask_hip_p2p.h, cask_amd/p2p.PushExchange) and the padded-stride layout of the
gathered vector: 1 to 5 processes share the box's one GPU (gloo control plane), each with its own HIP context, its
own shared region and mappings of every peer's -- the kernel, the address tables, the flags and the double-buffered
gathered vectors are exactly what runs with one GPU per rank.  (What one GPU cannot show is a stale line in another
GPU's L2: see DESIGN.md section 7.)"""
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
        kind = case["matrix"]
        n, rp, ci, va = synth.GENERATORS[kind[1]]() if kind[0] == "full" else synth.small(kind[1], factor=kind[2])
        inplace = case["exchange"] == "push_inplace"           # the slice written where it lives in the gathered vector
        sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, exchange="push" if inplace else case["exchange"])
        b0, b1 = sh.bounds[rank], sh.bounds[rank + 1]
        rng = np.random.default_rng(77)
        ys, gathered_ok = [], True
        y = torch.zeros(b1 - b0, dtype=torch.float64, device="cuda")
        for k in range(case["products"]):
            # the operand changes with every product: a slice that arrives one exchange late shows
            x = rng.uniform(-1, 1, n) + k
            if inplace:
                slot = sh.push.own_slot()                        # alternates with the gathered vectors
                slot[: b1 - b0].copy_(torch.from_numpy(x[b0:b1]).cuda())
                xf = sh.push.allgather(slot)                     # no own copy: stores to the peers only
                assert xf.data_ptr() + 8 * rank * sh.S == slot.data_ptr()
            else:
                sh.x_slot[: b1 - b0].copy_(torch.from_numpy(x[b0:b1]).cuda())
                if case["exchange"] == "all_gather":             # host-staged collective: order the streams by hand
                    torch.cuda.synchronize()
                xf = sh.gather_x(sh.x_slot)
            sh.local_product(xf, y)
            if k in (0, case["products"] - 1):
                torch.cuda.synchronize()
                gathered_ok = gathered_ok and bool(np.array_equal(sh.unpad(xf).cpu().numpy(), x))
                ys.append((k, y.cpu().numpy().copy()))
        torch.cuda.synchronize()
        if case["exchange"].startswith("push"):
            sh.push.check()
        out.put((rank, {"bounds": (b0, b1), "ys": ys, "gathered_ok": gathered_ok, "S": sh.S}))
        dist.barrier()
        sh.close()
    finally:
        dist.destroy_process_group()


def run_world(world, case):
    return spawn_collect(_worker, (world, free_port(), case), world)


def check(res, matrix, products):
    n, rp, ci, va = synth.GENERATORS[matrix[1]]() if matrix[0] == "full" else synth.small(matrix[1], factor=matrix[2])
    rng = np.random.default_rng(77)
    xs = [rng.uniform(-1, 1, n) + k for k in range(products)]
    for r in res:
        assert r["gathered_ok"]
    for idx, k in enumerate((0, products - 1)):
        got = np.concatenate([r["ys"][idx][1] for r in res])
        oracle.assert_almost_equal(got, oracle.csr_spmv(rp, ci, va, xs[k]), what=f"product {k}")


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_push_allgather_uneven_blocks_changing_operand(world):
    """nnz-balanced (uneven) row blocks of the power-law family, 12 chained products each on a different x: every
    rank's gathered vector (unpadded) equals the global x bit for bit, every product matches the oracle."""
    matrix = ("small", "webbase-1M", 16)
    res = run_world(world, {"matrix": matrix, "exchange": "push", "products": 12})
    if world > 1:
        assert len({r["bounds"][1] - r["bounds"][0] for r in res}) > 1          # genuinely uneven
    assert res[0]["S"] % 32 == 0
    check(res, matrix, 12)


def test_push_allgather_slice_in_place():
    """cask_hip_push_own_slot: the producer writes its slice where it lives in the gathered vector (it alternates with
    the two vectors); the exchange then only stores to the peers."""
    matrix = ("small", "webbase-1M", 16)
    res = run_world(3, {"matrix": matrix, "exchange": "push_inplace", "products": 9})
    check(res, matrix, 9)


def test_padded_stride_allgather_is_one_collective():
    """The same with the collective (torch.distributed all_gather_into_tensor of S doubles: no pad / copy kernels)."""
    matrix = ("small", "webbase-1M", 16)
    res = run_world(3, {"matrix": matrix, "exchange": "all_gather", "products": 4})
    check(res, matrix, 4)


def test_full_size_webbase_in_five_blocks_push():
    """BASELINE configs[3] at full size, 5 row blocks by 5 processes, slices pushed peer to peer."""
    matrix = ("full", "webbase-1M")
    res = run_world(5, {"matrix": matrix, "exchange": "push", "products": 3})
    check(res, matrix, 3)


def _reduce_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cask_amd import p2p
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def gather_objects(o):
            res = [None] * world
            dist.all_gather_object(res, o)
            return res
        ex = p2p.PushExchange(rank, world, 2, torch.device("cuda", 0), gather_objects)
        rng = np.random.default_rng(100 + rank)
        got = []
        for k in range(40):                                    # chained reductions of 1..4 values, values change every time
            cnt = 1 + k % 4
            t = torch.from_numpy(rng.standard_normal(cnt) * 10.0 ** (k % 7)).cuda()
            ex.allreduce(t)
            got.append(t.cpu().numpy().copy())
        torch.cuda.synchronize()
        ex.check()
        out.put((rank, got))
        dist.barrier()
        for p in ex.peers.values():
            p2p.close_peer(p)
        ex.peers = {}
        dist.barrier()
        ex.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 3, 5])
def test_peer_store_allreduce_same_bits_on_every_rank(world):
    """cask_hip_push_allreduce: 40 chained reductions; every rank ends with the SAME bits (rank-order sum), equal to
    the rank-order sum computed on the host."""
    out = spawn_collect(_reduce_worker, (world, free_port()), world)
    rngs = [np.random.default_rng(100 + r) for r in range(world)]
    for k in range(40):
        cnt = 1 + k % 4
        parts = [rng.standard_normal(cnt) * 10.0 ** (k % 7) for rng in rngs]
        want = np.zeros(cnt)
        for p in parts:                                        # rank order
            want = want + p
        for r in range(world):
            assert np.array_equal(out[r][k], want), (k, r)
