"""Seeded synthetic stand-ins for the SuiteSparse matrices BASELINE.json names.

The real files (cant, G3_circuit, webbase-1M, atmosmodd) are not in the
reference checkout and there is no network, so each generator reproduces the
published shape statistics: dimension, nnz, row-length
profile, bandedness, symmetry.  If ``$CASK_MATRIX_DIR/<name>.mtx`` exists the
benchmark uses the real file instead (see ``load_or_make``).

All generators return ``(n, row_ptr[int32], col_ind[int32], values[float64])``
with sorted, duplicate-free rows, and are pure numpy so they run identically
in the build container and on the GPU box.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

import hashlib as _hashlib

_SOURCE_HASH = _hashlib.sha1(Path(__file__).read_bytes()).hexdigest()   # part of every cache key (see _disk_cached)

SPECS = {
    # name: (n, nnz of the real SuiteSparse matrix)
    "cant": (62_451, 4_007_383),
    "G3_circuit": (1_585_478, 7_660_826),
    "webbase-1M": (1_000_005, 3_105_536),
    "webbase2": (1_000_005, 3_105_536),       # the same SuiteSparse matrix, second look-alike (webbase2_like)
    "atmosmodd": (1_270_432, 8_814_880),
}


def _disk_cached(fn):
    """Generated matrices are deterministic functions of their arguments and take 2-5 s each at full size; a test suite
    or a multi-rank run generates the same one in a dozen processes.  Full-size results are kept as .npy files under
    $CASK_SYNTH_CACHE (default: <tmp>/cask_synth_cache; "0" = off), keyed by generator, arguments and a hash of this
    file (an edited generator never meets a stale matrix); written under a temporary name and renamed, so concurrent
    ranks see whole files or none.  Small instances (< 200 K nonzeros) are not worth a file."""
    import functools
    import hashlib
    import tempfile

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        root = os.environ.get("CASK_SYNTH_CACHE", "")
        if root == "0":
            return fn(*args, **kwargs)
        key = hashlib.sha1((fn.__name__ + repr(args) + repr(sorted(kwargs.items())) + _SOURCE_HASH).encode()).hexdigest()[:20]
        # per user, mode 0700 (r6, ADVICE r5: a shared, predictable <tmp>/cask_synth_cache let another user of the box plant a
        # matrix under the right key)
        d = Path(root or os.path.join(tempfile.gettempdir(), f"cask_synth_cache_{os.getuid()}"))
        stem = d / f"{fn.__name__}_{key}"
        try:
            private = bool(root) or (d.exists() and d.stat().st_uid == os.getuid() and not (d.stat().st_mode & 0o077))
            if private and (d / f"{stem.name}.done").exists():
                n = int(np.load(f"{stem}_n.npy"))
                return n, np.load(f"{stem}_rp.npy"), np.load(f"{stem}_ci.npy"), np.load(f"{stem}_va.npy")
        except Exception:  # noqa: BLE001 - a damaged cache entry: regenerate
            pass
        n, rp, ci, va = fn(*args, **kwargs)
        if ci.size >= 200_000:
            try:
                d.mkdir(parents=True, exist_ok=True, mode=0o700)
                if not root and (d.stat().st_uid != os.getuid() or (d.stat().st_mode & 0o077)):
                    raise OSError("cache directory is not private to this user")
                for tag, arr in (("n", np.asarray(n)), ("rp", rp), ("ci", ci), ("va", va)):
                    tmp = f"{stem}_{tag}.{os.getpid()}.tmp.npy"
                    np.save(tmp, arr)
                    os.replace(tmp, f"{stem}_{tag}.npy")
                (d / f"{stem.name}.done").write_text("ok")
            except OSError:
                pass
        return n, rp, ci, va
    return wrapper


def _coo_to_csr(n, rows, cols, vals):
    """Sort by (row, col), keep the first of any duplicate."""
    key = rows.astype(np.int64) * n + cols.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    first = np.ones(key.size, dtype=bool)
    first[1:] = key[1:] != key[:-1]
    key, vals = key[first], vals[first]
    r = (key // n).astype(np.int64)
    c = (key % n).astype(np.int32)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_ptr, r + 1, 1)
    return np.cumsum(row_ptr).astype(np.int32), c, vals.astype(np.float64)


@_disk_cached
def cant_like(n=62_451, per_row=33, band=400, seed=0):
    """Symmetric banded FEM-like SPD matrix, ~2*per_row+1 nnz per row (cant: ~64)."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n, dtype=np.int64), per_row)
    cols = rows + rng.integers(1, band + 1, size=rows.size)
    vals = rng.standard_normal(rows.size)
    keep = cols < n
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    # unique upper-triangle pairs, then mirror
    key = rows * n + cols
    _, idx = np.unique(key, return_index=True)
    rows, cols, vals = rows[idx], cols[idx], vals[idx]
    absum = np.zeros(n)
    np.add.at(absum, rows, np.abs(vals))
    np.add.at(absum, cols, np.abs(vals))
    diag = np.arange(n, dtype=np.int64)
    R = np.concatenate([rows, cols, diag])
    C = np.concatenate([cols, rows, diag])
    V = np.concatenate([vals, vals, 1.0 + absum])
    rp, ci, va = _coo_to_csr(n, R, C, V)
    return n, rp, ci, va


@_disk_cached
def g3_like(n=1_585_478, nx=1259, extra_frac=0.01, seed=2):
    """2-D 5-point grid Laplacian + 1% random long-range symmetric edges, shifted SPD (~4.8 nnz/row)."""
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.int64)
    right = i[(i % nx != nx - 1) & (i + 1 < n)]
    down = i[i + nx < n]
    n_extra = int(extra_frac * n)
    ea = rng.integers(0, n, size=n_extra)
    eb = rng.integers(0, n, size=n_extra)
    ok = ea != eb
    ea, eb = ea[ok], eb[ok]
    a = np.concatenate([right, down, np.minimum(ea, eb)])
    b = np.concatenate([right + 1, down + nx, np.maximum(ea, eb)])
    key = a * n + b
    _, idx = np.unique(key, return_index=True)
    a, b = a[idx], b[idx]
    deg = np.zeros(n)
    np.add.at(deg, a, 1.0)
    np.add.at(deg, b, 1.0)
    R = np.concatenate([a, b, i])
    C = np.concatenate([b, a, i])
    V = np.concatenate([-np.ones(a.size), -np.ones(a.size), deg + 1e-3])
    rp, ci, va = _coo_to_csr(n, R, C, V)
    return n, rp, ci, va


@_disk_cached
def webbase_like(n=1_000_005, nnz_target=3_105_536, alpha=2.1, max_row=4700, seed=3):
    """Power-law row lengths (Zipf alpha, clipped), 70% of columns within +-1000 of the diagonal."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.zipf(alpha, size=n), max_row).astype(np.int64)
    # rescale towards the target nnz while keeping every row >= 1 and the tail
    scale = nnz_target / lens.sum()
    lens = np.clip(np.rint(lens * scale), 1, max_row).astype(np.int64)
    lens[rng.integers(0, n)] = max_row                       # make sure the longest row exists
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    local = rng.random(rows.size) < 0.7
    cols = np.where(local, rows + rng.integers(-1000, 1001, size=rows.size),
                    rng.integers(0, n, size=rows.size))
    cols = np.clip(cols, 0, n - 1)
    vals = rng.random(rows.size)
    rp, ci, va = _coo_to_csr(n, rows, cols, vals)
    return n, rp, ci, va


@_disk_cached
def webbase2_like(n=1_000_005, nnz_target=3_105_536, alpha=2.1, max_row=4700, seed=7, oversample=1.113):
    """Second webbase-1M look-alike: the same power-law OUT-degrees as ``webbase_like``, but the
    columns of a web graph instead of uniformly random ones --

    * pages are grouped into SITES (runs of consecutive ids, sizes power-law distributed, 8 .. 20 000 pages);
    * ~70 % of a page's links stay inside its site (within +-1000 of the page, clipped to the site);
    * the other ~30 % go to OTHER sites: every site has a handful of favourite external sites (drawn by site popularity,
      Zipf) that most of its pages' external links go to (site-block locality of the far columns), the rest go to
      globally popular sites; inside the target site the page is drawn from a Zipf over its pages (the home page first):
      power-law IN-degree, hub columns.

    Same n, nnz within a few per cent (duplicate links merge), same longest row."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.zipf(alpha, size=n), max_row).astype(np.int64)
    scale = nnz_target * oversample / lens.sum()
    lens = np.clip(np.rint(lens * scale), 1, max_row).astype(np.int64)
    lens[rng.integers(0, n)] = max_row
    # sites
    sizes = []
    total = 0
    while total < n:
        chunk = np.clip(rng.zipf(1.6, size=4096) * 8, 8, 20_000)
        sizes.append(chunk)
        total += int(chunk.sum())
    sizes = np.concatenate(sizes)
    ends = np.cumsum(sizes)
    n_sites = int(np.searchsorted(ends, n, side="left")) + 1
    sizes, ends = sizes[:n_sites].copy(), ends[:n_sites].copy()
    ends[-1] = n
    starts = np.concatenate([[0], ends[:-1]])
    sizes = ends - starts
    site_of_row = np.repeat(np.arange(n_sites), sizes)
    # site popularity (Zipf over a random order of the sites) and every site's favourite external sites
    rank_of_site = rng.permutation(n_sites)
    pop = 1.0 / (1.0 + rank_of_site) ** 1.1
    cdf = np.cumsum(pop) / pop.sum()
    n_fav = 4
    fav = np.searchsorted(cdf, rng.random((n_sites, n_fav))).clip(0, n_sites - 1)

    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    site = site_of_row[rows]
    u = rng.random(rows.size)
    local = u < 0.7
    # (a page with more links than its neighbourhood has pages -- a directory -- reaches proportionally further, and
    # not only into its own site: the longest rows keep their length instead of collapsing onto duplicates)
    reach = np.maximum(1000, 2 * lens)[rows]
    off = np.rint((rng.random(rows.size) * 2.0 - 1.0) * reach).astype(np.int64)
    wide = lens[rows] > 200
    near = np.where(wide, np.clip(rows + off, 0, n - 1), np.clip(rows + off, starts[site], ends[site] - 1))
    # external links: 80 % to one of the site's favourites, 20 % to a globally popular site
    pick_fav = rng.random(rows.size) < 0.8
    target = np.where(pick_fav, fav[site, rng.integers(0, n_fav, size=rows.size)],
                      np.searchsorted(cdf, rng.random(rows.size)).clip(0, n_sites - 1))
    # page inside the target site: Zipf over its pages, wrapped into the site
    page = (rng.zipf(1.5, size=rows.size) - 1) % sizes[target]
    far = starts[target] + page
    cols = np.where(local, near, far)
    vals = rng.random(rows.size)
    rp, ci, va = _coo_to_csr(n, rows, cols, vals)
    return n, rp, ci, va


@_disk_cached
def atmosmodd_like(n=1_270_432, nx=108, ny=108, seed=4):
    """Nonsymmetric 3-D 7-point advection-diffusion stencil (~6.9 nnz/row), diagonally dominant."""
    del seed
    i = np.arange(n, dtype=np.int64)
    plane = nx * ny
    parts_r, parts_c, parts_v = [i], [i], [np.full(n, 6.5)]

    def add(mask, off, coef):
        src = i[mask]
        parts_r.append(src)
        parts_c.append(src + off)
        parts_v.append(np.full(src.size, coef))

    add((i % nx != nx - 1) & (i + 1 < n), 1, -1.0 - 0.2)
    add(i % nx != 0, -1, -1.0 + 0.2)
    add(((i // nx) % ny != ny - 1) & (i + nx < n), nx, -1.0 - 0.1)
    add((i // nx) % ny != 0, -nx, -1.0 + 0.1)
    add(i + plane < n, plane, -1.0 - 0.05)
    add(i - plane >= 0, -plane, -1.0 + 0.05)
    rp, ci, va = _coo_to_csr(n, np.concatenate(parts_r), np.concatenate(parts_c), np.concatenate(parts_v))
    return n, rp, ci, va


@_disk_cached
def cant3_like(nx=9, ny=9, nz=257, order="z_fastest", seed=6):
    """A second cant look-alike, closer to the real FEM matrix: a 9 x 9 x 257 hexahedral mesh of a
    cantilever beam (20 817 nodes x 3 dof = 62 451 rows, the real dimension), 27-point node stencil, every node
    pair coupled by a DENSE 3 x 3 block (~69 nnz per row, 4.3 M nnz; the real cant: 64 and 4.0 M), symmetric
    positive definite (random symmetric blocks + diagonal dominance).  ``order="z_fastest"`` numbers the nodes
    along the beam first: the band is then wide and non-uniform (node offsets +-1, +-257 and +-2313 in
    clusters: three bands 7.7 K columns apart) -- not the one +-400 window of ``cant_like``;
    ``"x_fastest"`` gives the narrow band (+-273 columns)."""
    rng = np.random.default_rng(seed)
    n_nodes = nx * ny * nz
    X, Y, Z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    X, Y, Z = X.ravel(), Y.ravel(), Z.ravel()

    def node_id(x, y, z):
        return (x * ny + y) * nz + z if order == "z_fastest" else x + nx * (y + ny * z)

    me = node_id(X, Y, Z)
    pa, pb = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if (dx, dy, dz) <= (0, 0, 0):
                    continue                                   # each undirected pair once
                ok = (X + dx >= 0) & (X + dx < nx) & (Y + dy >= 0) & (Y + dy < ny) & (Z + dz >= 0) & (Z + dz < nz)
                pa.append(me[ok])
                pb.append(node_id(X[ok] + dx, Y[ok] + dy, Z[ok] + dz))
    pa, pb = np.concatenate(pa), np.concatenate(pb)
    # off-diagonal node pairs: a dense 3x3 block B at (a, b) and B^T at (b, a)
    blk = rng.standard_normal((pa.size, 3, 3))
    i3, j3 = np.meshgrid(np.arange(3), np.arange(3), indexing="ij")
    rows_ab = (3 * pa[:, None, None] + i3).ravel()
    cols_ab = (3 * pb[:, None, None] + j3).ravel()
    vals_ab = blk.ravel()
    # diagonal node blocks: symmetric, strictly dominant
    n = 3 * n_nodes
    absum = np.zeros(n)
    np.add.at(absum, rows_ab, np.abs(vals_ab))
    np.add.at(absum, cols_ab, np.abs(vals_ab))
    dblk = rng.standard_normal((n_nodes, 3, 3))
    dblk = 0.5 * (dblk + dblk.transpose(0, 2, 1))
    nodes = np.arange(n_nodes)
    rows_d = (3 * nodes[:, None, None] + i3).ravel()
    cols_d = (3 * nodes[:, None, None] + j3).ravel()
    vals_d = dblk.ravel().copy()
    offd = np.abs(dblk).sum(axis=2) - np.abs(dblk[:, [0, 1, 2], [0, 1, 2]])
    on_diag = rows_d == cols_d
    vals_d[on_diag] = 1.0 + absum + offd.ravel()
    R = np.concatenate([rows_ab, cols_ab, rows_d])
    C = np.concatenate([cols_ab, rows_ab, cols_d])
    V = np.concatenate([vals_ab, vals_ab, vals_d])
    rp, ci, va = _coo_to_csr(n, R, C, V)
    return n, rp, ci, va


def cant3_like_shard(rank, world):
    """cant3_like as a bench workload: one GPU only (the weak-scaling seam construction exists for cant_like)."""
    if world != 1:
        raise ValueError("the cant3 workload is defined for one GPU")
    n, rp, ci, va = cant3_like()
    return n, n, rp, ci, va


GENERATORS = {"cant": cant_like, "G3_circuit": g3_like, "webbase-1M": webbase_like, "webbase2": webbase2_like,
              "atmosmodd": atmosmodd_like}


def small(name, factor=16):
    """A reduced-size instance of the same family (for parity tests)."""
    n_full = SPECS[name][0]
    n = max(64, n_full // factor)
    if name == "cant":
        return cant_like(n=n)
    if name == "G3_circuit":
        return g3_like(n=n, nx=max(8, int(round((n) ** 0.5))))
    if name == "webbase-1M":
        return webbase_like(n=n, nnz_target=int(SPECS[name][1] / factor), max_row=min(4700, n // 2))
    if name == "webbase2":
        return webbase2_like(n=n, nnz_target=int(SPECS[name][1] / factor), max_row=min(4700, n // 2))
    if name == "atmosmodd":
        side = max(4, int(round(n ** (1 / 3))))
        return atmosmodd_like(n=n, nx=side, ny=side)
    raise KeyError(name)


def row_stats(row_ptr):
    lens = np.diff(row_ptr)
    return {"n": int(lens.size), "nnz": int(row_ptr[-1]), "row_min": int(lens.min()) if lens.size else 0,
            "row_mean": float(lens.mean()) if lens.size else 0.0, "row_max": int(lens.max()) if lens.size else 0,
            "row_std": float(lens.std()) if lens.size else 0.0, "empty_rows": int((lens == 0).sum())}


def algorithmic_bytes(n_rows, n_cols, nnz):
    """fp64 value + int32 column per nonzero, row_ptr, x read once, y written once."""
    return 12 * nnz + 4 * (n_rows + 1) + 8 * n_cols + 8 * n_rows


def load_or_make(name):
    """Real SuiteSparse file from $CASK_MATRIX_DIR if present, else the synthetic look-alike.
    Returns (n, row_ptr, col_ind, values, source)."""
    d = os.environ.get("CASK_MATRIX_DIR")
    if d:
        p = Path(d) / f"{name}.mtx"
        if p.exists():
            # the product's own reader (cask::io::readMatrixCached: symmetric expansion, binary cache), not scipy's
            from . import hostio
            n, m, rp, ci, va = hostio.read_matrix(p, cached=True)
            if n != m:
                raise ValueError(f"{p}: the BASELINE workloads are square, this file is {n} x {m}")
            return n, rp, ci, va, str(p)
    shrink = int(os.environ.get("CASK_BENCH_SHRINK", "0") or 0)
    if shrink > 1:                                          # tests of PROTOCOL (fault injection, fallbacks), where size is not the point
        n, rp, ci, va = small(name, factor=shrink)
        return n, rp, ci, va, f"synthetic, 1/{shrink} of the rows (CASK_BENCH_SHRINK)"
    n, rp, ci, va = GENERATORS[name]()
    return n, rp, ci, va, "synthetic"


def cant_like_shard(rank, world, n_local=62_451, per_row=33, band=400, n_couple=4000):
    """Row block `rank` of a (world*n_local)-square block-banded SPD matrix for weak scaling:
    diagonal blocks are independent cant_like() instances (seed = rank), neighbouring blocks are
    coupled through the band that crosses their seam, so the product needs the neighbours' x.
    Columns are GLOBAL indices.  Returns (n_local, n_global, row_ptr, col_ind, values)."""
    n, rp, ci, va = cant_like(n=n_local, per_row=per_row, band=band, seed=rank)
    n_global = world * n_local
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    cols = ci.astype(np.int64) + rank * n_local
    vals = va.copy()
    extra_r, extra_c, extra_v = [], [], []

    def seam(g):
        """coupling between block g (last rows) and block g+1 (first columns): (i_in_g, j_in_g1, v)."""
        rng = np.random.default_rng(10_000 + g)
        i = rng.integers(n_local - band, n_local, size=n_couple)
        off = rng.integers(1, band + 1, size=n_couple)
        j = i + off - n_local
        ok = j >= 0
        i, j = i[ok], j[ok]
        key = i * band + j
        _, idx = np.unique(key, return_index=True)
        return i[idx], j[idx], 0.01 * rng.standard_normal(idx.size)

    diag_add = np.zeros(n)
    if rank + 1 < world:
        i, j, v = seam(rank)
        extra_r.append(i); extra_c.append(j + (rank + 1) * n_local); extra_v.append(v)
        np.add.at(diag_add, i, np.abs(v))
    if rank > 0:
        i, j, v = seam(rank - 1)                      # transpose side: row j of this block, column i of the previous
        extra_r.append(j); extra_c.append(i + (rank - 1) * n_local); extra_v.append(v)
        np.add.at(diag_add, j, np.abs(v))
    if extra_r:
        rows = np.concatenate([rows] + extra_r)
        cols = np.concatenate([cols] + extra_c)
        vals = np.concatenate([vals] + extra_v)
    is_diag = cols == rows + rank * n_local
    vals = vals + np.where(is_diag, diag_add[rows], 0.0)
    key = rows * n_global + cols
    order = np.argsort(key, kind="stable")
    rows, cols, vals = rows[order], cols[order], vals[order]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    return n, n_global, np.cumsum(row_ptr).astype(np.int32), cols.astype(np.int32), vals
