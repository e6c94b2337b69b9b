"""Host-side plan of the peer-to-peer exchange (cask_amd/p2p.py): extended column indices, halo
ownership and the address table.  Pure numpy -- the device half is tests/test_p2p_gpu.py."""
import re
from pathlib import Path

import numpy as np
import pytest

import oracle
from cask_amd import capi, p2p, synth
from cask_amd import dist as cdist

REPO = Path(__file__).resolve().parent.parent


def simulate(n, rp, ci, va, x, bounds):
    """What the device path computes, rank by rank, with the oracle as the local product."""
    world = len(bounds) - 1
    y = np.empty(n)
    halos = []
    for g in range(world):
        lrp, lci, lva = cdist.slice_rows(rp, ci, va, bounds[g], bounds[g + 1])
        ci_ext, cols, owner, index = p2p.plan_halo(lci, bounds, g)
        n_local = bounds[g + 1] - bounds[g]
        # the pull: entry j comes from slice `owner[j]` at position `index[j]`
        halo = np.array([x[bounds[o] + i] for o, i in zip(owner, index)], dtype=np.float64)
        x_ext = np.concatenate([x[bounds[g]:bounds[g + 1]], halo])
        assert ci_ext.size == 0 or (ci_ext.min() >= 0 and ci_ext.max() < n_local + cols.size)
        y[bounds[g]:bounds[g + 1]] = oracle.csr_spmv(lrp, ci_ext, lva, x_ext)
        halos.append((cols, owner, index))
    return y, halos


@pytest.mark.parametrize("world", [2, 3, 8])
def test_extended_columns_reproduce_the_global_product(world):
    n, rp, ci, va = synth.webbase_like(n=700, nnz_target=4000, max_row=120, seed=world)
    x = np.random.default_rng(1).uniform(-1, 1, n)
    bounds = cdist.partition_rows_by_nnz(rp, world)
    y, halos = simulate(n, rp, ci, va, x, bounds)
    assert np.array_equal(y, oracle.csr_spmv(rp, ci, va, x))        # same products in the same order: bit-exact
    for g, (cols, owner, index) in enumerate(halos):
        assert np.all(owner != g)
        assert np.all(np.diff(cols) > 0)
        assert np.array_equal(np.asarray(bounds)[owner] + index, cols)


def test_banded_blocks_have_a_small_halo():
    world = 4
    blocks = [synth.cant_like_shard(g, world, n_local=2000, per_row=9) for g in range(world)]
    n_local, n_global = blocks[0][0], blocks[0][1]
    bounds = [g * n_local for g in range(world + 1)]
    for g, (_, _, rp, ci, va) in enumerate(blocks):
        ci_ext, cols, owner, index = p2p.plan_halo(ci, bounds, g)
        assert set(owner.tolist()) <= {g - 1, g + 1}                 # seams couple neighbours only
        assert 0 < cols.size <= 2 * 400                              # the band, not the 3*n_local of an all-gather
        assert ci_ext.max() < n_local + cols.size


def test_no_remote_columns():
    ci_ext, cols, owner, index = p2p.plan_halo(np.array([0, 3, 2], dtype=np.int32), [0, 4, 8], 0)
    assert cols.size == 0 and owner.size == 0 and ci_ext.tolist() == [0, 3, 2]
    ci_ext, cols, owner, index = p2p.plan_halo(np.array([], dtype=np.int32), [0, 4, 8], 1)
    assert ci_ext.size == 0 and cols.size == 0


def test_column_outside_partition_is_rejected():
    with pytest.raises(ValueError):
        p2p.plan_halo(np.array([9], dtype=np.int32), [0, 4, 8], 0)


def test_address_table():
    bases = [0x7f0000000000, 0, 0x7e0000001000]
    addr = p2p.halo_addresses([0, 2, 2], [5, 0, 7], bases)
    assert addr.dtype == np.int64
    assert addr.view(np.uint64).tolist() == [bases[0] + 40, bases[2], bases[2] + 56]


def test_header_and_binding_agree():
    """include/cask_hip_p2p.h declares exactly what p2p.py binds, and the library exports it."""
    text = re.sub(r"/\*.*?\*/", "", (REPO / "include" / "cask_hip_p2p.h").read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(cask_hip_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(p2p.P2P_SYMBOLS)
    lib = capi.load()
    for s in declared:
        assert hasattr(lib, s), s
