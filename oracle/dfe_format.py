"""The reference host's DFE stream format and call sequence -- TEST INFRASTRUCTURE ONLY.

Restates, in numpy, what src/runtime/Spmv.cpp does between a CsrMatrix and the device function
triple, so tests can drive the triple exactly the way the reference's unchanged Spmv.cpp would:

  preprocess      Spmv.cpp:329-365  rows split over num_pipes (n / num_pipes each, remainder last;
                                    n < num_pipes: one real partition + zero-valued copies)
  do_blocking     Spmv.cpp:42-107   column stripes of cache_size (SparseMatrix.hpp:459-482), per
                                    stripe n cumulative row ends, records padded to a multiple of
                                    input_width, packed {double,int32} 12-byte records (Spmv.hpp:14-20)
  write / run / read  Spmv.cpp:144-183, 185-328   384-byte burst padding, LMem address chain
                                    records | x | colptr | out per partition, offsets restarting per controller
"""
from __future__ import annotations

import numpy as np

BURST = 384
RECORD = np.dtype([("value", "<f8"), ("index", "<i4")])       # packed, 12 bytes
assert RECORD.itemsize == 12


def _pad_to(arr, multiple_bytes):
    """utils::align (Utils.hpp:62-69): append zero elements until the byte size is a multiple."""
    item = arr.dtype.itemsize
    per = multiple_bytes // item
    if per == 0:
        return arr
    rem = (arr.size * item) % multiple_bytes
    if rem == 0:
        return arr
    add = min(-(-(multiple_bytes - rem) // item), per)
    return np.concatenate([arr, np.zeros(add, dtype=arr.dtype)])


def encode_empty_rows(row_ends):
    """SkipEmptyRowsSpmv::encodeEmptyRows (Spmv.hpp:213-237): a run of k empty rows becomes one entry k | 1<<31,
    a non-empty row keeps its cumulative end."""
    out, run, prev = [], 0, 0
    for e in row_ends.tolist():
        if e == prev:
            run += 1
        else:
            if run:
                out.append(run | (1 << 31))
            run = 0
            out.append(e)
        prev = e
    if run:
        out.append(run | (1 << 31))
    return np.asarray(out, dtype=np.uint32).view(np.int32)


def do_blocking(n_cols, rp, ci, va, cache_size, input_width, rle=False):
    """One partition (a row slice with n rows): returns dict(colptr, records, n, n_blocks, out_bytes, vlc).
    rle: the SkipEmptyRowsSpmv architecture -- the column pointers of every block but the first and the last are
    run-length encoded (preprocessBlock, Spmv.hpp:240-250)."""
    n = rp.size - 1
    n_blocks = n_cols // cache_size + (0 if n_cols % cache_size == 0 else 1)
    rows = np.repeat(np.arange(n), np.diff(rp))
    blk = ci // cache_size
    colptr_parts, rec_parts = [], []
    for b in range(n_blocks):
        sel = blk == b
        cnt = np.bincount(rows[sel], minlength=n)
        ends = np.cumsum(cnt).astype(np.int32)                          # n cumulative row ENDS, no leading 0
        colptr_parts.append(encode_empty_rows(ends) if rle and 0 < b < n_blocks - 1 else ends)
        rec = np.zeros(int(sel.sum()), dtype=RECORD)
        rec["value"] = va[sel]
        rec["index"] = ci[sel] - b * cache_size
        rec_parts.append(_pad_to(rec, 12 * input_width))
    v_len = _pad_to(np.zeros(n_cols), 8 * cache_size).size
    out_len = _pad_to(np.zeros(n), BURST).size
    return {"colptr": np.concatenate(colptr_parts) if colptr_parts else np.zeros(0, np.int32),
            "records": np.concatenate(rec_parts) if rec_parts else np.zeros(0, RECORD),
            "n": n, "n_blocks": n_blocks, "out_bytes": out_len * 8, "vector_load_cycles": v_len // max(n_blocks, 1)}


def preprocess(n_rows, n_cols, rp, ci, va, num_pipes, cache_size, input_width, rle=False):
    per = n_rows // num_pipes
    if per == 0:
        p = do_blocking(n_cols, rp, ci, va, cache_size, input_width, rle)
        z = dict(p)
        z["records"] = p["records"].copy()
        z["records"]["value"] = 0
        return [p] + [z] * (num_pipes - 1)
    parts, start = [], 0
    for i in range(num_pipes):
        cnt = per if i < num_pipes - 1 else n_rows - start
        k0, k1 = rp[start], rp[start + cnt]
        parts.append(do_blocking(n_cols, rp[start:start + cnt + 1] - k0, ci[k0:k1], va[k0:k1], cache_size, input_width, rle))
        start += cnt
    return parts


def spmv_through_triple(triple, n_rows, parts, x, num_pipes, num_controllers, cache_size):
    """Spmv::spmv (Spmv.cpp:185-328) against a device function triple.
    triple = (write(size, sizes[], starts[], bytes, routing), run(...12 args...), read(size, sizes[], starts[], nbytes, routing) -> bytes)."""
    write, run, read = triple
    if len(parts) != num_pipes:
        raise RuntimeError("numPipes should equal numPartitions")
    if num_pipes % num_controllers != 0:
        raise RuntimeError("numPipes should be a multiple of numControllers")
    v = _pad_to(_pad_to(np.asarray(x, dtype=np.float64), 8 * cache_size), BURST)
    per_ctrl = num_pipes // num_controllers
    cols = {k: [] for k in ("colptr_start", "colptr_size", "rec_start", "rec_size", "nrows", "out_start", "out_size",
                            "v_start", "reduction", "total")}
    offset = 0

    def one_hot(ctrl, value):
        a = np.zeros(num_controllers, dtype=np.int64)
        a[ctrl] = value
        return a

    def write_padded(ctrl, start, arr):
        data = _pad_to(arr, BURST)
        nbytes = data.size * data.dtype.itemsize
        write(nbytes, one_hot(ctrl, nbytes), one_hot(ctrl, start), data.tobytes(), f"split -> tomem{ctrl}")
        return nbytes

    for i, p in enumerate(parts):
        ctrl = i // per_ctrl
        if i % per_ctrl == 0:
            offset = 0
        rec_start = -(-offset // BURST) * BURST
        rec_bytes = write_padded(ctrl, rec_start, p["records"])
        v_start = rec_start + rec_bytes
        v_bytes = write_padded(ctrl, v_start, v)
        cp_start = v_start + v_bytes
        cp_bytes = write_padded(ctrl, cp_start, p["colptr"])
        out_start = cp_start + cp_bytes
        cols["colptr_start"].append(cp_start)
        cols["colptr_size"].append(p["colptr"].size * 4)
        cols["rec_start"].append(rec_start)
        cols["rec_size"].append(p["records"].size * 12)
        cols["nrows"].append(p["n"])
        cols["out_start"].append(out_start)
        cols["out_size"].append(p["out_bytes"])
        cols["v_start"].append(v_start)
        cols["reduction"].append(p["n"] * p["n_blocks"])
        cols["total"].append(0)
        offset = out_start + p["out_bytes"]
    i64 = lambda k: np.asarray(cols[k], dtype=np.int64)      # noqa: E731
    i32 = lambda k: np.asarray(cols[k], dtype=np.int32)      # noqa: E731
    run(2, parts[0]["n_blocks"], parts[0]["vector_load_cycles"], i64("colptr_start"), i32("colptr_size"),
        i64("rec_start"), i32("rec_size"), i32("nrows"), i64("out_start"), i32("reduction"), i32("total"), i64("v_start"))
    total = []
    for i, p in enumerate(parts):
        ctrl = i // per_ctrl
        raw = read(cols["out_size"][i], one_hot(ctrl, cols["out_size"][i]), one_hot(ctrl, cols["out_start"][i]),
                   cols["out_size"][i], f"frommem{ctrl} -> join")
        total.append(np.frombuffer(raw, dtype=np.float64)[: p["n"]])
    y = np.concatenate(total) if total else np.zeros(0)
    return y[:n_rows] if n_rows < num_pipes else y
