"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle and the committed golden vectors.  Tolerance is the reference's own
(test/test_utils.hpp:36): relative 1e-8 / absolute 1e-11 -- summation order on
the GPU (butterfly) differs from the sequential order of the oracle, so the
comparison is never bitwise; run-to-run results of one design point ARE bitwise
identical (no float atomics) and that is tested too."""
import numpy as np
import pytest

import oracle
from oracle import mmio
from cask_amd import capi, synth
from conftest import golden_matrix_files

pytestmark = pytest.mark.gpu

# design points that exercise every kernel family / template branch
DESIGN_POINTS = [
    dict(variant="vector", lanes_per_row=1, tile_width=-1),
    dict(variant="vector", lanes_per_row=4, tile_width=-1, wg_size=64),
    dict(variant="vector", lanes_per_row=16, tile_width=4096),
    dict(variant="vector", lanes_per_row=64, tile_width=-1, nontemporal=-1),
    dict(variant="vector", lanes_per_row=32, tile_width=512, xcd_remap=-1, wg_size=512),
    dict(variant="merge", items_per_thread=2, wg_size=64, tile_width=-1),
    dict(variant="merge", items_per_thread=8, wg_size=256, tile_width=4096),
    dict(variant="merge", items_per_thread=4, wg_size=128, tile_width=64, xcd_remap=-1),
    dict(variant="merge", items_per_thread=16, wg_size=256, tile_width=-1, nontemporal=-1),
    dict(variant="merge", items_per_thread=8, wg_size=512, tile_width=1024),
    dict(variant="merge", items_per_thread=4, wg_size=256, tile_width=2048, index16=-1),
    dict(variant="merge", items_per_thread=8, wg_size=256, tile_width=4096, index16=2),       # 16-bit slots, not packed
    dict(variant="merge", items_per_thread=8, wg_size=128, tile_width=1024, index16=1, nontemporal=-1),   # 12-bit packed
    dict(variant="merge", items_per_thread=4, wg_size=512, tile_width=4096),                   # 16-bit slots (4 items per thread)
    dict(variant="merge", items_per_thread=8, wg_size=256, tile_width=2048),                   # 12-bit packed, the headline's shape
    dict(variant="merge", items_per_thread=2, wg_size=64, tile_width=128, nontemporal=-1),
    dict(variant="merge_wave", items_per_thread=2, wg_size=64),
    dict(variant="merge_wave", items_per_thread=4, wg_size=256, xcd_remap=-1),
    dict(variant="merge_wave", items_per_thread=8, wg_size=512, nontemporal=-1),
    dict(variant="merge_wave", items_per_thread=16, wg_size=128),
    dict(variant="scan", items_per_thread=8, wg_size=256),                                     # nonzero-mapped, segmented scan
    dict(variant="scan", items_per_thread=2, wg_size=64, nontemporal=-1),
    dict(variant="scan", items_per_thread=4, wg_size=128, xcd_remap=-1),
    dict(variant="scan", items_per_thread=16, wg_size=256),
    dict(variant="scan", items_per_thread=8, wg_size=256, tile_width=4096),                    # x window in LDS (the product area)
    dict(variant="scan", items_per_thread=4, wg_size=512, tile_width=2048),
    dict(variant="scan", items_per_thread=16, wg_size=128, tile_width=300, xcd_remap=-1),
    dict(variant="scan", items_per_thread=2, wg_size=64, tile_width=128, nontemporal=-1),
    dict(variant="slice"),                                                                     # r6: short rows row-mapped (K = 4), long rows SCAN blocks
    dict(variant="slice", lanes_per_row=1, wg_size=64, items_per_thread=4, tile_width=-1),
    dict(variant="slice", lanes_per_row=2, wg_size=128, items_per_thread=8, tile_width=512),
    dict(variant="slice", lanes_per_row=3, wg_size=256, items_per_thread=8, tile_width=2048),
    dict(variant="slice", lanes_per_row=8, wg_size=512, items_per_thread=4, tile_width=4096, xcd_remap=-1),
    dict(variant="slice", lanes_per_row=6, wg_size=1024, items_per_thread=4, tile_width=1024),
    dict(),   # AUTO / all defaults
]
DP_IDS = ["-".join(f"{k[:3]}{v}" for k, v in dp.items()) or "auto" for dp in DESIGN_POINTS]


def _covering_subset(points):
    """A subset of `points` that still holds every kernel family and, within a family, every value of every axis the
    full list holds (greedy set cover, deterministic).  The 43 reference fixtures are swept over this subset -- the axes
    select tile widths, load flavours, block mappings and index widths, and which COMBINATION of them meets a fixture's
    shape adds nothing once each has met it -- while the full list stays on the five synthetic families and on the
    random profiles (suite time)."""
    keys = ("variant", "lanes_per_row", "tile_width", "wg_size", "items_per_thread", "xcd_remap", "nontemporal", "index16", "far_columns")

    def cells(dp):
        return {(dp.get("variant", "auto"), k, dp.get(k, 0)) for k in keys}
    need = set().union(*(cells(dp) for dp in points))
    chosen = []
    while need:
        best = max(range(len(points)), key=lambda i: (len(cells(points[i]) & need), -i))
        chosen.append(best)
        need -= cells(points[best])
    return [points[i] for i in sorted(chosen)]


FIXTURE_POINTS = _covering_subset(DESIGN_POINTS)
FIXTURE_IDS = [DP_IDS[DESIGN_POINTS.index(dp)] for dp in FIXTURE_POINTS]


def run_host(n_rows, n_cols, rp, ci, va, x, dp):
    m = capi.CsrMatrix.from_host(n_rows, n_cols, rp, ci, va, capi.make_params(**dp))
    try:
        return m.spmv(x)
    finally:
        m.close()


@pytest.fixture(scope="module")
def fixtures_loaded():
    out = {}
    for key, path in golden_matrix_files():
        out[key] = mmio.read_matrix(path)
    return out


@pytest.mark.parametrize("dp", FIXTURE_POINTS, ids=FIXTURE_IDS)
def test_reference_fixtures_every_design_point(dp, fixtures_loaded, expected_y):
    """test/test_spmv.cpp protocol over all 43 fixture matrices (incl. the two 'failing' ones), for a covering subset of
    the design points (every kernel family; within a family every value of every axis)."""
    for key, m in fixtures_loaded.items():
        x = mmio.test_vector(m.m)
        got = run_host(m.n, m.m, m.row_ptr, m.col_ind, m.values, x, dp)
        oracle.assert_almost_equal(got, expected_y[key], what=f"{key} {dp}")


@pytest.mark.parametrize("name", list(synth.GENERATORS))
@pytest.mark.parametrize("dp", DESIGN_POINTS, ids=DP_IDS)
def test_baseline_families_small(name, dp):
    n, rp, ci, va = synth.small(name)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    got = run_host(n, n, rp, ci, va, x, dp)
    oracle.assert_almost_equal(got, want, what=f"{name} {dp}")


@pytest.mark.parametrize("name", list(synth.GENERATORS))
def test_baseline_configs_full_size(name):
    """BASELINE.json configs at full size: oracle comparison (the C oracle needs ~20 ms) plus
    size-independent properties: linearity and the column-sum checksum 1'(Ax) = (A'1)'x."""
    n, rp, ci, va = synth.GENERATORS[name]()
    rng = np.random.default_rng(12)
    x1, x2 = rng.uniform(-1, 1, n), mmio.test_vector(n) / n
    want1 = oracle.csr_spmv(rp, ci, va, x1)
    colsum = oracle.csr_spmv_t(n, rp, ci, va, np.ones(n))
    for dp in (dict(variant="merge"), dict(variant="vector"), dict(variant="merge", tile_width=-1),
               dict(variant="merge_wave"), dict(variant="merge", wg_size=512, items_per_thread=8),
               dict(variant="vector", lanes_per_row=8, tile_width=-1), dict(variant="scan"),
               dict(variant="scan", tile_width=-1), dict(variant="scan", items_per_thread=4, wg_size=512, tile_width=4096),
               dict(variant="slice", lanes_per_row=2 if name.startswith("webbase") else 8, tile_width=2048)):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
        y1, y2 = m.spmv(x1), m.spmv(x2)
        y12 = m.spmv(2.0 * x1 - 0.5 * x2)
        again = m.spmv(x1)
        m.close()
        oracle.assert_almost_equal(y1, want1, what=f"{name} {dp}")
        assert np.array_equal(y1, again), "SpMV must be bitwise reproducible run to run"
        np.testing.assert_allclose(y12, 2.0 * y1 - 0.5 * y2, rtol=1e-9, atol=1e-9 * np.abs(y1).max())
        np.testing.assert_allclose(y1.sum(), colsum @ x1, rtol=1e-9, atol=1e-8 * np.abs(y1).sum())


def _cant3_small():
    return synth.cant3_like(nx=9, ny=9, nz=33)                  # 8 019 rows, the full matrix's 3 x 3 node blocks


def test_removed_design_points_are_rejected_with_a_reason():
    """ABI 6 / 7: the measured losers are gone from the shipped engine -- variant MERGE_PAIR (5,
    xcd_remap = 2), run records (index16 = 3 / 4), the far-column pre-gathers (far_columns = 1 / 2: MERGE's in ABI 6, SCAN's
    in ABI 7) -- and asking for one is an error that names the replacement, at create and at set_params."""
    n, rp, ci, va = synth.small("cant", factor=16)
    assert capi.load().cask_hip_abi_version() >= 7
    for bad, word in ((dict(variant=capi.VARIANT_MERGE_PAIR_REMOVED), "MERGE_PAIR"), (dict(variant="merge", xcd_remap=2), "MERGE_PAIR"),
                      (dict(variant="merge", index16=4), "run records"), (dict(variant="merge", index16=3), "run records"),
                      (dict(variant="merge", far_columns=1), "were removed"), (dict(variant="vector", far_columns=2), "were removed"),
                      (dict(variant="scan", far_columns=1), "pre-gather"), (dict(variant="scan", far_columns=2), "pre-gather"),
                      (dict(variant="slice", lanes_per_row=9), "1..8"), (dict(variant="slice", items_per_thread=16), "4 or 8")):
        with pytest.raises(ValueError, match=word):
            capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**bad))
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    with pytest.raises(ValueError, match="run records"):
        m.set_params(capi.make_params(variant="merge", index16=4))
    x = mmio.test_vector(n)
    oracle.assert_almost_equal(m.spmv(x), oracle.csr_spmv(rp, ci, va, x), what="the handle keeps its plan after a rejected set_params")
    m.close()


def test_widest_window_shares_the_product_area():
    """r5: the 8-loads-per-lane x window of the >= 8-items merge kernels is parked in the product area's LDS (18.4 instead
    of 34.8 KB per 256 x 8 block: 8 instead of 4 workgroups per CU), the SCAN kernel's window always is.  Where x comes
    from does not change a single product or sum: bit-identical to the same plan with a narrower window of its own /
    without a window, and the handle reports the smaller LDS."""
    n, rp, ci, va = synth.atmosmodd_like()                      # full size: 7 narrow bands far apart, a chunked tile of ~2 000 slots
    x = np.random.default_rng(5).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    ys, lds = {}, {}
    for tile in (2048, -1):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="merge", wg_size=256, items_per_thread=8, tile_width=tile))
        ys[tile], lds[tile] = m.spmv(x), m.info.lds_bytes
        assert np.array_equal(ys[tile], m.spmv(x))
        if tile > 0:
            assert m.params.as_dict()["tile_width"] == 2048, m.params.as_dict()
        m.close()
        oracle.assert_almost_equal(ys[tile], want, what=f"merge tile {tile}")
    assert np.array_equal(ys[2048], ys[-1])
    assert lds[2048] == lds[-1] == 8 * (256 * 8 + 2) + 8 * 256  # products + row offsets: the window has no share of its own
    n, rp, ci, va = synth.small("webbase-1M", factor=8)
    x = np.random.default_rng(6).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    for ipt, wg in ((8, 256), (4, 512), (16, 128)):
        out = {}
        for tile in (4096, -1):
            m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="scan", wg_size=wg, items_per_thread=ipt, tile_width=tile))
            out[tile] = (m.spmv(x), m.info.lds_bytes, m.params.as_dict()["tile_width"])
            m.close()
            oracle.assert_almost_equal(out[tile][0], want, what=f"scan {ipt} x {wg} tile {tile}")
        assert np.array_equal(out[4096][0], out[-1][0])
        assert out[4096][1] == out[-1][1] and 0 < out[4096][2] <= ipt * wg, out


def test_edge_shapes():
    # no rows at all
    m = capi.CsrMatrix.from_host(0, 0, [0], [], [])
    assert m.spmv(np.zeros(0)).size == 0
    m.close()
    # rows but no nonzeros (test_large_empty-like)
    for dp in (dict(variant="merge"), dict(variant="vector", lanes_per_row=2), dict(variant="scan")):
        y = run_host(1000, 7, np.zeros(1001, dtype=np.int32), [], [], np.ones(7), dp)
        assert y.shape == (1000,) and not y.any()
    # rectangular, wider than tall and taller than wide
    rng = np.random.default_rng(2)
    import scipy.sparse as sp
    for shape in ((50, 3000), (3000, 50)):
        a = sp.random(*shape, 0.05, format="csr", random_state=rng)
        a.sort_indices()
        x = rng.standard_normal(shape[1])
        want = oracle.csr_spmv(a.indptr, a.indices, a.data, x)
        for dp in DESIGN_POINTS:
            got = run_host(shape[0], shape[1], a.indptr, a.indices, a.data, x, dp)
            oracle.assert_almost_equal(got, want, what=f"rect {shape} {dp}")


def test_long_rows_split_across_workgroups():
    """One row far longer than a workgroup's share (webbase-like tail): long-row pieces + fix-up."""
    rng = np.random.default_rng(4)
    n = 3000
    lens = np.full(n, 3)
    lens[17] = 2900          # > 16*CAP for CAP=128 -> split into pieces
    lens[18] = 0
    lens[1500] = 700         # long but a single piece
    lens[n - 1] = 1500
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int32)
    va = rng.standard_normal(ci.size)
    x = rng.standard_normal(n)
    want = oracle.csr_spmv(rp, ci, va, x)
    for dp in (dict(variant="merge", wg_size=64, items_per_thread=2),
               dict(variant="merge", wg_size=64, items_per_thread=2, tile_width=256),
               dict(variant="merge", wg_size=256, items_per_thread=8),
               dict(variant="merge_wave", items_per_thread=2), dict(variant="merge_wave", items_per_thread=16, wg_size=64),
               dict(variant="vector", lanes_per_row=64), dict(variant="vector", lanes_per_row=1),
               dict(variant="scan", wg_size=64, items_per_thread=2), dict(variant="scan", wg_size=256, items_per_thread=8),
               dict(variant="scan", wg_size=64, items_per_thread=4, tile_width=512),
               dict(variant="slice", wg_size=64, items_per_thread=4), dict(variant="slice", lanes_per_row=8, wg_size=256, tile_width=512)):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
        info = m.info
        got = m.spmv(x)
        m.close()
        oracle.assert_almost_equal(got, want, what=f"long rows {dp}")
        if dp["variant"] == "merge" and dp["wg_size"] == 64:
            assert info.n_long_rows == 3 and info.n_split_rows >= 1
        if dp["variant"] == "merge_wave":
            assert info.n_long_rows >= 1


def test_device_vector_entry_point_and_streams():
    import torch
    n, rp, ci, va = synth.small("cant", factor=8)
    x = mmio.test_vector(n)
    want = oracle.csr_spmv(rp, ci, va, x)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    xt = torch.from_numpy(x).cuda()
    yt = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    m.spmv_device(xt, yt)                              # torch's current stream
    torch.cuda.synchronize()
    oracle.assert_almost_equal(yt.cpu().numpy(), want, what="device entry")
    side = torch.cuda.Stream()
    yt2 = torch.zeros_like(yt)
    with torch.cuda.stream(side):
        m.spmv_device(xt, yt2, stream=side)
    side.synchronize()
    assert torch.equal(yt, yt2)
    m.close()


def test_borrowed_device_arrays():
    import torch
    n, rp, ci, va = synth.small("atmosmodd", factor=8)
    x = mmio.test_vector(n) / n
    want = oracle.csr_spmv(rp, ci, va, x)
    rpt, cit, vat = (torch.from_numpy(a).cuda() for a in (rp, ci, va))
    m = capi.CsrMatrix.from_device(n, n, rpt, cit, vat, capi.make_params(variant="merge"))
    xt = torch.from_numpy(x).cuda()
    yt = torch.empty(n, dtype=torch.float64, device="cuda")
    m.spmv_device(xt, yt)
    torch.cuda.synchronize()
    oracle.assert_almost_equal(yt.cpu().numpy(), want, what="borrowed arrays")
    m.close()


def test_transpose_product():
    import torch
    n, rp, ci, va = synth.small("webbase-1M", factor=8)
    x = np.random.default_rng(9).uniform(-1, 1, n)
    want = oracle.csr_spmv_t(n, rp, ci, va, x)
    for dp in (dict(), dict(variant="scan", tile_width=2048), dict(variant="merge", tile_width=2048, wg_size=256)):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
        xt = torch.from_numpy(x).cuda()
        yt = torch.empty(n, dtype=torch.float64, device="cuda")
        m.spmv_transpose_device(xt, yt)          # the transpose handle is planned with the same design point
        torch.cuda.synchronize()
        oracle.assert_almost_equal(yt.cpu().numpy(), want, what=f"A^T x {dp}")
        m.close()


def test_size_mismatch_raises_like_the_reference():
    """Spmv::spmv throws std::invalid_argument on shape violations (Spmv.cpp:189-207)."""
    m = capi.CsrMatrix.from_host(3, 3, [0, 1, 2, 3], [0, 1, 2], [1.0, 1.0, 1.0])
    with pytest.raises(ValueError):
        m.spmv(np.ones(4))
    with pytest.raises(ValueError):
        m.set_params(capi.make_params(variant="vector", lanes_per_row=3))
    with pytest.raises(ValueError):
        m.set_params(capi.make_params(variant="merge", wg_size=1024, items_per_thread=16))
    assert np.array_equal(m.spmv(np.array([1.0, 2.0, 3.0])), [1.0, 2.0, 3.0])   # handle still usable
    m.close()


def test_tune_sweeps_in_reference_order_and_keeps_the_best(monkeypatch):
    monkeypatch.setenv("CASK_HIP_TUNE_NO_PRUNE", "1")          # every point measured (pruning: the test below)
    n, rp, ci, va = synth.small("cant", factor=8)
    x = mmio.test_vector(n)
    want = oracle.csr_spmv(rp, ci, va, x)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    pts, best = m.tune(variants=[capi.VARIANT_VECTOR, capi.VARIANT_MERGE, capi.VARIANT_MERGE_WAVE], lanes=[4, 16, 64],
                       tiles=[-1, 2048], wg_sizes=[256], items=[4, 8], warmup=1, iters=5)
    # first list (variants) fastest; lanes only swept for VECTOR, items only for the merge kernels,
    # tiles not for MERGE_WAVE
    assert len(pts) == 3 * 2 + 2 * 2 + 2
    assert pts[0]["params"]["variant"] == "vector" and pts[0]["params"]["lanes_per_row"] == 4
    assert pts[1]["params"]["variant"] == "merge"
    assert all(p["valid"] and p["usec"] > 0 for p in pts)
    assert pts[best]["usec"] == min(p["usec"] for p in pts)
    assert m.params.as_dict() == pts[best]["params"]
    oracle.assert_almost_equal(m.spmv(x), want, what="after tune")
    m.close()


def test_tune_prunes_a_family_that_is_far_behind():
    """a family is dropped only after the two points its PRIOR ranks best are both > 1.5x
    behind the incumbent (VECTOR: the lane counts nearest half the mean row length -- here 8 and 4 of [1, 2, 4, 8] on
    64-nonzero rows -- not the first two in odometer order, which are its worst); its remaining points come back as
    valid = 0 / usec = -1 and the order of the list is the odometer's."""
    n, rp, ci, va = synth.small("cant", factor=4)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    pts, best = m.tune(variants=[capi.VARIANT_VECTOR, capi.VARIANT_MERGE], lanes=[1, 2, 4, 8], tiles=[-1], wg_sizes=[256], items=[8])
    assert len(pts) == 4 + 1
    vec = [p for p in pts if p["params"]["variant"] == "vector"]
    assert [p["params"]["lanes_per_row"] for p in vec] == [1, 2, 4, 8]            # reported in the odometer's order
    assert [p["valid"] for p in vec[2:]] == [True, True]        # 4 and 8 lanes per 64-nonzero row: measured first
    assert not any(p["valid"] for p in vec[:2]) and all(p["usec"] == -1.0 for p in vec[:2])   # a thread / two lanes: never
    assert pts[best]["params"]["variant"] == "merge"
    m.close()


def test_tune_measures_the_lane_count_nearest_the_row_length_first():
    """BASELINE configs[1] is "DSE over rows-per-wavefront" -- on 64-nonzero rows the sweep must reach
    L = 32 (2 rows per wavefront), the row-mapped family's best, before it may drop the family."""
    n, rp, ci, va = synth.small("cant", factor=4)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    pts, best = m.tune(variants=[capi.VARIANT_VECTOR, capi.VARIANT_MERGE], lanes=[4, 8, 16, 32], tiles=[-1, 1024],
                       wg_sizes=[256], items=[8])
    vec = {(p["params"]["lanes_per_row"], p["params"]["tile_width"]): p for p in pts if p["params"]["variant"] == "vector"}
    measured = [k for k, p in vec.items() if p["valid"]]
    assert any(k[0] == 32 for k in measured), measured
    l32 = min(p["usec"] for k, p in vec.items() if k[0] == 32 and p["valid"])
    others = [p["usec"] for k, p in vec.items() if k[0] in (4, 8) and p["valid"]]
    assert all(l32 < t for t in others)                        # and it is the family's best where both were measured
    m.close()


def test_tune_keeps_a_family_that_is_within_reach():
    """A family whose best guesses are within 1.5x of the incumbent keeps ALL its points: the row-mapped family against
    the persistent-wave merge kernel alone (cant-like: 13.4 vs 12.1 us at full size) -- neither may lose a point.
    (On this chip the row-mapped family is never within 1.5x of the workgroup-level merge kernel: a scaled-up
    test_dense_128 -- 2048 rows of 128 consecutive nonzeros, timed warm -- measures 3.7 against 2.4 us.)"""
    n, rp, ci, va = synth.small("cant", factor=4)
    x = mmio.test_vector(n)
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    pts, best = m.tune(variants=[capi.VARIANT_VECTOR, capi.VARIANT_MERGE_WAVE], lanes=[4, 8, 16, 32], tiles=[-1, 1024],
                       wg_sizes=[256], items=[4, 8])
    top = min(p["usec"] for p in pts if p["valid"])
    for fam in ("vector", "merge_wave"):
        mine = [p for p in pts if p["params"]["variant"] == fam]
        two_best = sorted(p["usec"] for p in mine if p["valid"])[:2]
        if min(two_best) <= 1.5 * top:
            assert all(p["valid"] and p["usec"] > 0 for p in mine), (fam, [(p["params"]["lanes_per_row"], p["usec"]) for p in mine])
    fams = {p["params"]["variant"] for p in pts if p["valid"]}
    assert fams == {"vector", "merge_wave"}
    oracle.assert_almost_equal(m.spmv(x), oracle.csr_spmv(rp, ci, va, x), what="after tune")
    m.close()


def test_one_dse_cold_and_warm_times():
    """ONE DSE.  cask_hip_tune ranks every point on its COLD time (launches rotate over device
    copies that exceed 2x the Infinity Cache) and records the warm time next to it; cask_amd.dse.explore -- what
    bench.py and tools/dse.py call -- is that same call, and build/main makes it too (tests/test_host_cpp.py)."""
    from cask_amd import dse
    n, rp, ci, va = synth.small("cant", factor=4)             # 12 MB of matrix: an HBM workload, 40+ copies
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va)
    pts, best = m.tune(variants=[capi.VARIANT_MERGE], tiles=[-1, 1024], wg_sizes=[256], items=[8])
    assert len(pts) == 2 and all(p["valid"] and p["copies"] > 8 for p in pts)
    assert all(p["usec"] > 0 and p["usec_warm"] > 0 and p["usec"] >= 0.9 * p["usec_warm"] for p in pts)
    assert pts[best]["usec"] == min(p["usec"] for p in pts)
    rows, best_row, _ = dse.explore([m], points=[dict(variant="merge", items_per_thread=8, tile_width=t, wg_size=256)
                                                  for t in (-1, 1024)])
    assert len(rows) == 2 and {"usec", "usec_warm", "copies"} <= set(rows[0])
    assert m.params.as_dict()["tile_width"] == best_row["tile_width"]
    m.close()
    # a matrix that fits the L2 is not an HBM workload: timed warm only
    n2, rp2, ci2, va2 = synth.small("cant", factor=64)
    m = capi.CsrMatrix.from_host(n2, n2, rp2, ci2, va2)
    pts, _ = m.tune(variants=[capi.VARIANT_MERGE], tiles=[1024], wg_sizes=[256], items=[8])
    assert pts[0]["copies"] == 1 and pts[0]["usec"] == pts[0]["usec_warm"]
    m.close()


def _random_csr(rng, n_rows, n_cols, profile):
    """Seeded random CSR with a chosen row-length profile (sorted, duplicate-free rows)."""
    if profile == "uniform":
        lens = rng.integers(0, 12, n_rows)
    elif profile == "powerlaw":
        lens = np.minimum(rng.zipf(1.8, n_rows), n_cols)
    elif profile == "banded":
        lens = np.full(n_rows, min(24, n_cols))
    elif profile == "mostly_empty":
        lens = np.where(rng.random(n_rows) < 0.05, rng.integers(1, 40, n_rows), 0)
    else:   # "dense_rows"
        lens = np.full(n_rows, min(n_cols, 200))
    lens = np.minimum(lens, n_cols).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = []
    for r, l in enumerate(lens):
        if profile == "banded":
            lo = max(0, min(n_cols - l, r * n_cols // max(n_rows, 1) - l // 2))
            cols.append(np.arange(lo, lo + l))
        else:
            cols.append(np.sort(rng.choice(n_cols, size=l, replace=False)))
    ci = (np.concatenate(cols) if cols else np.zeros(0)).astype(np.int32)
    va = rng.standard_normal(ci.size)
    return rp, ci, va


@pytest.mark.parametrize("profile", ["uniform", "powerlaw", "banded", "mostly_empty", "dense_rows"])
def test_random_matrices_all_design_points(profile):
    """Property test: for seeded random shapes and row-length profiles every design point agrees with the
    oracle under the reference tolerance, and odd/even nnz, odd block starts and ragged tails are hit."""
    rng = np.random.default_rng({"uniform": 1, "powerlaw": 2, "banded": 3, "mostly_empty": 4, "dense_rows": 5}[profile])
    for trial in range(6):
        n_rows = int(rng.integers(1, 3000))
        n_cols = int(rng.integers(1, 3000))
        rp, ci, va = _random_csr(rng, n_rows, n_cols, profile)
        x = rng.uniform(-2, 2, n_cols)
        want = oracle.csr_spmv(rp, ci, va, x)
        for dp in DESIGN_POINTS:
            got = run_host(n_rows, n_cols, rp, ci, va, x, dp)
            oracle.assert_almost_equal(got, want, what=f"{profile} trial {trial} {n_rows}x{n_cols} nnz={ci.size} {dp}")


def test_empty_row_block_at_odd_end_of_arrays():
    """a block of only empty rows whose (clamped) stream pair straddles the end of col_ind used to
    feed the uninitialised int behind the array to an x gather.  col_ind sits at the front of a larger device
    allocation whose next element is poison (a huge column): a wild gather would fault or return garbage."""
    import torch
    cases = []
    # (a) 2 nnz in row 0, a long run of empty rows, 1 trailing nonzero: nnz odd, an all-empty block starts at the
    #     even index nnz-1
    for n_empty in (600, 1021, 2500):
        n = n_empty + 2
        rp = np.zeros(n + 1, dtype=np.int32)
        rp[1:] = 2
        rp[n] = 3
        cases.append((n, rp, np.array([0, 1, 5], dtype=np.int32), np.array([1.0, 2.0, 3.0])))
    # (b) odd nnz followed by trailing empty rows: the all-empty block starts at nnz itself
    for n_empty in (600, 2500):
        n = n_empty + 3
        rp = np.zeros(n + 1, dtype=np.int32)
        rp[1], rp[2], rp[3:] = 1, 2, 3
        cases.append((n, rp, np.array([0, 1, 2], dtype=np.int32), np.array([1.0, 2.0, 3.0])))
    for n, rp, ci, va in cases:
        x = np.arange(1, n + 1, dtype=np.float64)
        want = oracle.csr_spmv(rp, ci, va, x)
        for dp in (dict(variant="merge", tile_width=-1), dict(variant="merge"), dict(variant="merge", wg_size=64, items_per_thread=2),
                   dict(variant="merge_wave"), dict(variant="merge_wave", wg_size=64, items_per_thread=2)):
            big = torch.full((ci.size + 61,), 0x7ffffff0, dtype=torch.int32, device="cuda")
            big[: ci.size] = torch.from_numpy(ci).cuda()
            vbig = torch.full((va.size + 61,), float("nan"), dtype=torch.float64, device="cuda")
            vbig[: va.size] = torch.from_numpy(va).cuda()
            m = capi.CsrMatrix.from_device(n, n, torch.from_numpy(rp).cuda(), big[: ci.size], vbig[: va.size],
                                           capi.make_params(**dp))
            xt = torch.from_numpy(x).cuda()
            yt = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
            m.spmv_device(xt, yt)
            torch.cuda.synchronize()
            m.close()
            oracle.assert_almost_equal(yt.cpu().numpy(), want, what=f"empty-block n={n} {dp}")


def test_create_device_rejects_out_of_range_columns():
    """borrowed device arrays are range-checked at create time (a wild column is an OOB gather)."""
    import torch
    rp = torch.tensor([0, 2, 3], dtype=torch.int32, device="cuda")
    va = torch.ones(3, dtype=torch.float64, device="cuda")
    for bad in ([0, 5, 1], [-1, 0, 1]):
        ci = torch.tensor(bad, dtype=torch.int32, device="cuda")
        for dp in (dict(variant="vector"), dict(variant="merge", tile_width=-1), dict()):
            with pytest.raises(ValueError, match="column index out of range"):
                capi.CsrMatrix.from_device(2, 3, rp, ci, va, capi.make_params(**dp))


def test_cant3_fem_blocks_wide_band():
    """the second cant look-alike -- 3x3 dense node blocks on a 9 x 9 x 257 beam mesh numbered so
    that the band is wide and non-uniform (three bands 7.7 K columns apart): parity under the design points of the
    headline kernel, and what its plan looks like (12-bit packed slots still apply; the tile is no longer ONE window)."""
    n, rp, ci, va = synth.cant3_like()
    assert n == 62_451 and 4_200_000 < ci.size < 4_450_000
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    colsum = oracle.csr_spmv_t(n, rp, ci, va, np.ones(n))
    for dp in (dict(), dict(variant="merge", tile_width=1024), dict(variant="merge", tile_width=4096, wg_size=512),
               dict(variant="merge", tile_width=-1), dict(variant="vector", lanes_per_row=16), dict(variant="merge_wave")):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(**dp))
        y, again = m.spmv(x), m.spmv(x)
        prm = m.params.as_dict()
        m.close()
        oracle.assert_almost_equal(y, want, what=f"cant3 {dp}")
        assert np.array_equal(y, again)
        np.testing.assert_allclose(y.sum(), colsum @ x, rtol=1e-9, atol=1e-8 * np.abs(y).sum())
        if not dp:
            assert prm["variant"] == "merge" and prm["index16"] == 1 and prm["tile_width"] > 0   # 12-bit slots, tiled
    # the narrow-band numbering of the same mesh
    n2, rp2, ci2, va2 = synth.cant3_like(order="x_fastest")
    x2 = rng.uniform(-1, 1, n2)
    m = capi.CsrMatrix.from_host(n2, n2, rp2, ci2, va2)
    oracle.assert_almost_equal(m.spmv(x2), oracle.csr_spmv(rp2, ci2, va2, x2), what="cant3 x_fastest")
    m.close()


def test_sequence_of_products_from_one_call():
    """cask_hip_spmv_sequence_device: k products in stream order, rotating over handles of one shape (what bench.py
    times for short regions); the last product's y is the last handle's."""
    import torch
    n, rp, ci, va = synth.small("cant", factor=32)
    x = np.random.default_rng(5).uniform(-1, 1, n)
    mats = [capi.CsrMatrix.from_host(n, n, rp, ci, va * s) for s in (1.0, 2.0, 3.0)]
    xt = torch.from_numpy(x).cuda()
    yt = torch.zeros(n, dtype=torch.float64, device="cuda")
    base = oracle.csr_spmv(rp, ci, va, x)
    for k, scale in ((1, 1.0), (5, 2.0), (9, 3.0)):             # product k-1 uses handle (k-1) % 3
        capi.spmv_sequence_device(mats, xt, yt, k)
        torch.cuda.synchronize()
        oracle.assert_almost_equal(yt.cpu().numpy(), scale * base, what=f"sequence k={k}")
    other = capi.CsrMatrix.from_host(4, 4, [0, 1, 2, 3, 4], [0, 1, 2, 3], [1.0] * 4)
    with pytest.raises(ValueError):
        capi.spmv_sequence_device([mats[0], other], xt, yt, 2)
    for m in mats + [other]:
        m.close()


@pytest.mark.parametrize("as_graph", [False, True], ids=["stream", "one-graph"])
def test_timed_windows_from_one_call(as_graph):
    """cask_hip_spmv_windows_device (r4): `windows` back-to-back groups of k products, each between two timing events --
    as stream launches, or as the kernel nodes of ONE graph with event-record nodes at the window boundaries; returns
    every window's microseconds after the last has completed.  The last product's y is handle (windows*k - 1) % 3's."""
    import torch
    n, rp, ci, va = synth.small("cant", factor=32)
    x = np.random.default_rng(5).uniform(-1, 1, n)
    mats = [capi.CsrMatrix.from_host(n, n, rp, ci, va * s) for s in (1.0, 2.0, 3.0)]
    xt = torch.from_numpy(x).cuda()
    yt = torch.zeros(n, dtype=torch.float64, device="cuda")
    base = oracle.csr_spmv(rp, ci, va, x)
    us = capi.spmv_windows_device(mats, xt, yt, 5, 7, as_graph=as_graph)           # 35 products: the last uses handle 34 % 3 = 1
    assert us.shape == (7,) and np.all(us > 0) and np.all(us < 1e5)
    oracle.assert_almost_equal(yt.cpu().numpy(), 2.0 * base, what="windows: the last product")
    assert us.max() < 20 * us.min()                              # windows of equal work
    with pytest.raises(ValueError):
        capi.spmv_windows_device(mats, xt, yt, 0, 3, as_graph=as_graph)
    capi.spmv_windows_device(mats, xt, yt, 2, 2, as_graph=False)                    # the handle and the stream stay usable
    torch.cuda.synchronize()
    for m in mats:
        m.close()


def test_scan_kernel_rows_that_span_threads_waves_and_holes():
    """The segmented-scan kernel (variant SCAN) where its carries matter: rows that span many threads and several
    waves, rows that end exactly on a thread's / a wave's last product, 1-nonzero rows, empty rows between them
    (KIND_HOLES blocks: row map + zero fill), and a run of empty rows at either end.  Against the oracle, and
    bitwise reproducible."""
    rng = np.random.default_rng(31)
    for wg, ipt in ((64, 2), (256, 8), (128, 16)):
        cap = wg * ipt
        lens = []
        for _ in range(40):
            lens += [ipt] * 3 + [0] + [1] * (2 * ipt + 1) + [64 * ipt] + [0, 0] + [64 * ipt - 1, 1] + [3, 0, 5]
            lens += [int(v) for v in rng.integers(0, 4, size=50)] + [cap // 2, cap // 2 - 1, 1, cap // 4 + 3]
        lens = np.array([0] * 9 + lens + [0] * 5, dtype=np.int64)
        n = lens.size
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        m_cols = max(n, int(lens.max()) + 1)
        ci = np.concatenate([np.sort(rng.choice(m_cols, size=l, replace=False)) for l in lens]).astype(np.int32)
        va = rng.standard_normal(ci.size)
        x = rng.standard_normal(m_cols)
        want = oracle.csr_spmv(rp, ci, va, x)
        for extra in (dict(variant="scan", tile_width=-1), dict(variant="scan", tile_width=1024)) + \
                ((dict(variant="slice", lanes_per_row=3, tile_width=-1), dict(variant="slice", lanes_per_row=ipt, tile_width=512)) if ipt in (4, 8) else ()):
            m = capi.CsrMatrix.from_host(n, m_cols, rp, ci, va, capi.make_params(wg_size=wg, items_per_thread=ipt, **extra))
            assert m.params.as_dict()["variant"] == extra["variant"]
            got, again = m.spmv(x), m.spmv(x)
            m.close()
            oracle.assert_almost_equal(got, want, what=f"scan {wg}x{ipt} {extra}")
            assert np.array_equal(got, again)


def test_scan_window_is_taken_and_exact():
    """SCAN on the power-law family with its x window (LDS slots instead of columns): the same products in the same order
    whichever way x arrives => identical bits.  (The far-column pre-gathers of rounds 3-5 left the engine in ABI 7.)"""
    n, rp, ci, va = synth.webbase_like()
    x = np.random.default_rng(22).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    ys = {}
    for tile in (-1, 4096, 2048, 512):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="scan", tile_width=tile))
        prm = m.params.as_dict()
        ys[tile] = (m.spmv(x), prm["tile_width"], prm["far_columns"])
        assert np.array_equal(ys[tile][0], m.spmv(x))
        m.close()
        oracle.assert_almost_equal(ys[tile][0], want, what=f"scan tile {tile}")
    # (r5: the window lives in the product area's LDS, so it is at most wg_size * items_per_thread = 2 048 entries wide)
    assert ys[-1][1:] == (-1, -1) and ys[4096][1:] == (2048, -1) and ys[2048][1:] == (2048, -1) and ys[512][1:] == (1024, -1)
    for key in ys:
        assert np.array_equal(ys[key][0], ys[-1][0]), key


def test_slice_plan_shapes_and_bits():
    """Variant SLICE (r6) on the power-law look-alikes: a short row is added in stored order by ONE thread, so the rows
    of one nonzero carry the SAME BITS whatever K is; the plan reports K, the long rows' window, and a grid of nonzero-mapped
    + row-mapped workgroups; run to run bitwise reproducible; also as a captured graph with a changing operand."""
    import torch
    n, rp, ci, va = synth.small("webbase2", factor=4)            # 250 000 rows (the full size: test_baseline_configs_full_size)
    lens = np.diff(rp)
    x = np.random.default_rng(23).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    got = {}
    for k in (1, 3, 4, 8):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="slice", lanes_per_row=k, tile_width=2048))
        prm, info = m.params.as_dict(), m.info
        assert prm["variant"] == "slice" and prm["lanes_per_row"] == k and prm["tile_width"] == 2048
        rows_per_block = 512 if k <= 4 else 256
        assert info.grid >= (n + rows_per_block - 1) // rows_per_block
        got[k] = m.spmv(x)
        assert np.array_equal(got[k], m.spmv(x))
        m.close()
        oracle.assert_almost_equal(got[k], want, what=f"slice K={k}")
    short = lens <= 1
    for k in got:
        assert np.array_equal(got[k][short], got[1][short])    # a one-nonzero row is one product wherever it runs
    m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="slice", lanes_per_row=4))
    dev = torch.device("cuda", 0)
    xs = [torch.from_numpy(np.random.default_rng(40 + i).uniform(-1, 1, n)).to(dev) for i in range(4)]
    xt = torch.zeros(n, dtype=torch.float64, device=dev)
    yt = torch.zeros(n, dtype=torch.float64, device=dev)
    outs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(4)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.spmv_device(xt, yt)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(4):
            xt.copy_(xs[i])
            m.spmv_device(xt, yt)
            outs[i].copy_(yt)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for i in range(4):
        oracle.assert_almost_equal(outs[i].cpu().numpy(), oracle.csr_spmv(rp, ci, va, xs[i].cpu().numpy()), what=f"graph product {i}")
    m.close()


def test_vector_long_rows_take_the_long_row_path():
    """r6: the row-mapped VECTOR family hands rows of more than max(32, 16 L) nonzeros to long-row
    pieces (k_spmv_long + the fix-up for rows of several pieces), as the merge plans do -- it let L lanes walk a
    4 700-entry row before (260-300 us on the webbase look-alikes)."""
    n, rp, ci, va = synth.small("webbase-1M", factor=2)
    lens = np.diff(rp)
    x = np.random.default_rng(24).uniform(-1, 1, n)
    want = oracle.csr_spmv(rp, ci, va, x)
    for lanes in (1, 2, 8, 64):
        m = capi.CsrMatrix.from_host(n, n, rp, ci, va, capi.make_params(variant="vector", lanes_per_row=lanes, tile_width=-1))
        info = m.info
        assert info.n_long_rows == int((lens > max(32, 16 * lanes)).sum())
        assert info.n_split_rows == int((lens > 4096).sum())
        got = m.spmv(x)
        assert np.array_equal(got, m.spmv(x))
        m.close()
        oracle.assert_almost_equal(got, want, what=f"vector L={lanes}")
