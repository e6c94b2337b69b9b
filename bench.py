#!/usr/bin/env python3
"""Headline benchmark: fp64 CSR SpMV on the cant-like matrix (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path: y = A x over the whole matrix resident in
HBM.  For N > 1 (weak scaling: every rank owns one cant-sized row block of a
block-banded global matrix, and the slice of x that goes with it) a step is still
ONE launch per rank: the x slices live in shared allocations and the product
kernel's seam workgroups load the halo entries they need straight from the
neighbours' slices over xGMI (cask_hip_csr_set_halo_sources, cask_amd/p2p.py) --
no collective, no copy and no exchange kernel on the data path.  That mode is
verified bit-for-bit against the halo-pull mode (a small kernel in front of the
product) before it is timed; if the slices cannot be mapped the step falls back
to the pull, then to an RCCL all-gather of x.  To make the number an HBM number
and not an Infinity-Cache number the steps rotate through enough device copies of
the matrix to exceed 2x the 256 MiB cache ("cold"); the cache-warm rate of one
copy is reported next to it.  The K timed steps are captured once into a HIP
graph (one kernel node per step) so the host's launch rate is not what is
measured.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      -- algorithmic bytes per launch / mean launch time (HIP events on
                   the launch stream) against the 8 TB/s HBM peak
  cpu_baseline  -- the CPU oracle (1 core, "port") and MKL's dcsrgemv on all host
                   cores, both on the same matrix, bounded to ~10 s each.
"""
import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
INFINITY_CACHE_BYTES = 256 << 20


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cant", choices=["cant", "G3_circuit", "webbase-1M", "atmosmodd"])
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"])
    ap.add_argument("--no-tune", action="store_true", help="skip the measured DSE, use the AUTO design point")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--copies", type=int, default=0, help="matrix copies to rotate through (0 = auto)")
    ap.add_argument("--variant", default=None, help="force a design point: vector|merge")
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--wg", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(rp, ci, va, x, y_gpu, seconds):
    """Time the CPU oracle (sequential C restatement) and MKL on the same matrix."""
    import oracle
    out = {}
    n = rp.size - 1
    nnz = int(ci.size)
    y = oracle.csr_spmv(rp, ci, va, x)                  # warm-up + the checker
    bad, first = oracle.mismatches(y_gpu, y)
    t0 = time.perf_counter()
    calls = 0
    while True:
        oracle.csr_spmv(rp, ci, va, x)
        calls += 1
        el = time.perf_counter() - t0
        if el >= seconds or calls >= 5000:
            break
    out.update({"value": round(2.0 * nnz * calls / el / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                "sample": f"{calls} sequential CSR SpMVs of the same {n}x{n} matrix in {el:.1f} s (oracle/cask_oracle.c)",
                "parity_gpu_vs_cpu_mismatches": bad})
    # MKL, the CPU library the reference calls (fpgaNaiveCpuCode.cpp:33, SparseLinearSolvers.hpp:189)
    mkl = None
    for cand in (os.environ.get("MKLROOT", "/nonexistent") + "/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.1",
                 "/opt/conda/lib/libmkl_rt.so.2"):
        if os.path.exists(cand):
            try:
                mkl = ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
                break
            except OSError:
                pass
    if mkl is not None and x.size == n:
        try:
            mkl.MKL_Get_Max_Threads.restype = ctypes.c_int
            max_threads = int(mkl.MKL_Get_Max_Threads())
            tr, nn = ctypes.c_char(b"N"), ctypes.c_int(n)
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
            ym = np.zeros(n)
            args = (ctypes.byref(tr), ctypes.byref(nn), p(va), p(rp), p(ci), p(x), p(ym))
            # the host is shared and a 62 K-row product does not feed 128 threads: time a few team sizes for
            # a slice of the budget each and report the best one (its thread count is `threads`)
            counts = sorted({t for t in (8, 16, 32, 64, max_threads) if 0 < t <= max_threads})
            by_threads, best = {}, None
            for t in counts:
                mkl.MKL_Set_Num_Threads(ctypes.c_int(t))
                for _ in range(3):
                    mkl.mkl_cspblas_dcsrgemv(*args)
                t0 = time.perf_counter()
                calls = 0
                while True:
                    mkl.mkl_cspblas_dcsrgemv(*args)
                    calls += 1
                    el = time.perf_counter() - t0
                    if el >= seconds / len(counts) or calls >= 20000:
                        break
                rate = 2.0 * nnz * calls / el / 1e9
                by_threads[str(t)] = round(rate, 3)
                if best is None or rate > best[0]:
                    best = (rate, t, calls, el)
            mbad, _ = oracle.mismatches(ym, y)
            out["mkl"] = {"value": round(best[0], 4), "unit": "GFLOP/s", "threads": best[1],
                          "host_cores": os.cpu_count(), "routine": "mkl_cspblas_dcsrgemv",
                          "sample": f"{best[2]} calls in {best[3]:.1f} s", "gflops_by_threads": by_threads,
                          "mismatches_vs_oracle": mbad}
        except Exception as e:  # pragma: no cover - diagnostic only
            out["mkl"] = {"error": repr(e)}
    else:
        out["mkl"] = None
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # CASK_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, all_gather, all_reduce) with a single
    # rank: a dry run of the N>1 plumbing on a 1-GPU box.
    use_dist = world > 1 or bool(os.environ.get("CASK_BENCH_FORCE_DIST"))
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    os.environ.setdefault("OMP_PROC_BIND", "true")

    import torch
    import torch.distributed as dist
    from cask_amd import capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    backend = os.environ.get("CASK_BENCH_BACKEND", "nccl")
    if os.environ.get("CASK_BENCH_SHARE_DEVICE"):              # dry run of N ranks on a 1-GPU box (with the gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    ctl_dev = dev if backend == "nccl" else torch.device("cpu")

    def all_reduce_scalar(v, op):
        t = torch.tensor([float(v)], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=op)
        return float(t[0])

    def host_barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # ---- workload -----------------------------------------------------------
    if args.workload == "cant":
        n_local, n_global, rp, ci, va = synth.cant_like_shard(rank, world)
        source = "synthetic"
    else:
        if use_dist:
            raise SystemExit("multi-GPU bench is defined for the cant workload")
        n_local, rp, ci, va, source = synth.load_or_make(args.workload)
        n_global = n_local
    nnz_local = int(ci.size)
    x_host = np.arange(n_global, dtype=np.float64) * 0.25 / n_global        # test_spmv.cpp operand, scaled
    x_slice = x_host[rank * n_local:(rank + 1) * n_local].copy()

    # ---- exchange (N > 1): peer-to-peer halo pull, else RCCL all-gather ------------
    exchange, peer, p2p_error = "none", None, None
    ci_dev, n_cols_dev = ci, n_global
    if use_dist:
        exchange = "all_gather"
        if not os.environ.get("CASK_BENCH_NO_P2P"):
            from cask_amd import p2p
            bounds = [g * n_local for g in range(world + 1)]
            ci_ext, halo_cols, halo_owner, halo_index = p2p.plan_halo(ci, bounds, rank)

            def gather_objects(obj):
                out = [None] * world
                dist.all_gather_object(out, obj)
                return out

            try:
                peer = p2p.PeerExchange(bounds, rank, world, halo_owner, halo_index, dev, gather_objects)
                peer.x_local.copy_(torch.from_numpy(x_slice).to(dev))
                host_barrier()                                   # every slice is in place before anyone pulls
                peer.pull()
                torch.cuda.synchronize()
                # self-check against a halo computed from the formula for x (no collective involved)
                want = torch.from_numpy(x_host[halo_cols]).to(dev)
                ok = bool(torch.equal(peer.x_ext[n_local:], want))
            except Exception as e:  # noqa: BLE001 - setup is collective: it fails on every rank or none
                ok, p2p_error = False, repr(e)
            ok = all_reduce_scalar(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5
            if ok:
                exchange, ci_dev, n_cols_dev = "p2p", ci_ext, n_local + peer.n_halo
            else:
                if rank == 0:
                    print(f"[bench] peer-to-peer exchange unavailable ({p2p_error or 'halo mismatch'}); "
                          "using the RCCL all-gather", file=sys.stderr)
                if peer is not None:
                    peer.close()
                    peer = None
    # bytes the local kernel must move: x entries = the columns this block can reference
    alg_bytes = synth.algorithmic_bytes(n_local, n_cols_dev, nnz_local)
    matrix_bytes = 12 * nnz_local + 4 * (n_local + 1)
    copies = args.copies or max(2, -(-2 * INFINITY_CACHE_BYTES // matrix_bytes) + 1)

    forced = capi.make_params(variant=args.variant or 0, lanes_per_row=args.lanes, tile_width=args.tile,
                              items_per_thread=args.items, wg_size=args.wg,
                              index16=int(os.environ.get("CASK_BENCH_INDEX16", "0")))   # development A/B only
    mats = []
    rp_t = torch.from_numpy(rp).to(dev)
    for _ in range(copies):
        ci_t, va_t = torch.from_numpy(ci_dev).to(dev), torch.from_numpy(va).to(dev)
        mats.append(capi.CsrMatrix.from_device(n_local, n_cols_dev, rp_t, ci_t, va_t, forced))
    if exchange == "p2p":
        x_in = peer.x_ext                                        # [own slice | halo]
        if not os.environ.get("CASK_BENCH_NO_FUSED_HALO") and peer.n_halo:
            # Fold the exchange into the product kernel: the workgroups at a seam read the halo from the
            # peers' slices themselves (cask_hip_csr_set_halo_sources), a step is ONE launch.  Checked
            # against the pull path first; any rank that disagrees sends every rank back to the pull.
            y_ref = torch.zeros(n_local, dtype=torch.float64, device=dev)
            peer.pull()
            mats[0].spmv_device(x_in, y_ref)
            torch.cuda.synchronize()
            try:
                for m in mats:
                    peer.attach(m)
                y_try = torch.zeros(n_local, dtype=torch.float64, device=dev)
                peer.x_ext[n_local:].fill_(float("nan"))         # the kernel must not read the local halo copy
                mats[0].spmv_device(x_in, y_try)
                torch.cuda.synchronize()
                fused_ok = bool(torch.equal(y_try, y_ref))
            except Exception as e:  # noqa: BLE001
                fused_ok, p2p_error = False, repr(e)
            fused_ok = all_reduce_scalar(1.0 if fused_ok else 0.0, dist.ReduceOp.MIN) > 0.5
            if fused_ok:
                exchange = "p2p_fused"
            else:
                if rank == 0:
                    print(f"[bench] in-kernel halo disagrees with the pull path ({p2p_error}); pulling", file=sys.stderr)
                for m in mats:
                    m.set_halo_sources(n_local, None)
                peer.pull()
    elif exchange == "all_gather":
        x_local = torch.from_numpy(x_slice).to(dev)
        x_in = torch.zeros(n_global, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(x_in, x_local)
    else:
        x_in = torch.from_numpy(x_host).to(dev)
    y = torch.zeros(n_local, dtype=torch.float64, device=dev)

    # ---- measured DSE (cold: on the rotating copies), best point left active --------
    tune_info = None
    if not args.no_tune and args.variant is None:
        from cask_amd import dse
        rows, best, took = dse.explore(mats, x_in, y)
        if use_dist:
            # every rank runs the same design point: rank 0's winner
            obj = [mats[0].params.as_dict()]
            dist.broadcast_object_list(obj, src=0)
            prm = capi.make_params(**obj[0])
            for m in mats:
                m.set_params(prm)
        tune_info = {"points": len(rows), "best_usec_cold": best["usec"], "seconds": round(took, 2)}
    design = mats[0].params.as_dict()
    info = mats[0].info

    def step(i):
        if exchange == "p2p":
            peer.pull()                                          # remote loads over xGMI, on the launch stream
        elif exchange == "all_gather":
            dist.all_gather_into_tensor(x_in, x_local)
        mats[i % copies].spmv_device(x_in, y)

    # ---- warm-up (eager) -------------------------------------------------------
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    launch_mode = args.launch if exchange != "all_gather" else "eager"
    graph, preroll = None, 0
    if launch_mode == "graph":
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(3):
                    step(i)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for i in range(args.steps):
                    step(i)
            graph.replay()                                  # one untimed replay (graph upload)
            torch.cuda.synchronize()
            # The GPU leaves its idle power state only after ~10 ms of sustained load (tools/replay_series.py:
            # the first replays after a host-side pause run 3-4 % slower than the following ones), and the
            # capture above is such a pause.  Pre-roll ~40 ms of untimed replays so the K timed steps run at
            # the steady-state clock a solver loop sees.
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            preroll = int(min(200, max(2, 40.0 / max(e0.elapsed_time(e1), 1e-3))))
        except Exception as e:  # pragma: no cover
            print(f"[bench] graph capture failed ({e!r}); timing eager launches", file=sys.stderr)
            graph, launch_mode = None, "eager"
    if use_dist:
        # ranks must agree on the launch mode only for reporting; the timed region has no collective in p2p mode
        launch_mode = "graph" if all_reduce_scalar(1.0 if graph is not None else 0.0, dist.ReduceOp.MIN) > 0.5 \
            else "eager"
        if launch_mode == "eager":
            graph = None

    # ---- timed region: exactly K steps -------------------------------------------
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph is not None:
        for _ in range(preroll):
            graph.replay()
    host_barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    if graph is not None:
        graph.replay()
    else:
        for i in range(args.steps):
            step(i)
    e1.record()
    host_barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    y_gpu = y.cpu().numpy()
    if use_dist:
        elapsed = all_reduce_scalar(elapsed, dist.ReduceOp.MAX)
        nnz_total = all_reduce_scalar(nnz_local, dist.ReduceOp.SUM)
        # every rank checks its whole block against the CPU oracle (the global x is a formula, nothing to gather)
        import oracle
        bad, _ = oracle.mismatches(y_gpu, oracle.csr_spmv(rp, ci, va, x_host))
        rows_wrong = int(all_reduce_scalar(bad, dist.ReduceOp.SUM))
    else:
        nnz_total = float(nnz_local)
        rows_wrong = None

    # ---- cache-warm rate of ONE copy (what a CG iteration on this matrix sees) ------
    warm_med, warm_min = mats[0].time(x_in, y, warmup=10, iters=200)        # eager launches, one event pair each
    warm_graph = None
    if graph is not None:
        # the same thing measured the way `value` is: a graph of back-to-back launches of ONE copy
        try:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                for _ in range(200):
                    mats[0].spmv_device(x_in, y)
            g2.replay()
            torch.cuda.synchronize()
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record()
            g2.replay()
            w1.record()
            torch.cuda.synchronize()
            warm_graph = w0.elapsed_time(w1) * 1e3 / 200
        except Exception:  # pragma: no cover - reporting only
            warm_graph = None

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        gflops = 2.0 * nnz_total * args.steps / elapsed / 1e9
        launch_us = dev_ms * 1e3 / args.steps              # HIP events on the launch stream, whole timed region
        achieved = alg_bytes / (launch_us * 1e-6) / 1e9
        traffic = None
        tfile = REPO / "profiles" / f"traffic_{args.workload}.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        rec = {
            "metric": "SpMV GFLOP/s (fp64 CSR, 2*nnz/t), SuiteSparse cant-like", "value": round(gflops, 2),
            "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if source == "synthetic" else source,
            "config": {"workload": f"{args.workload}-like CSR SpMV, {n_local} rows x {n_global} cols per GPU, "
                                   f"{nnz_local} nnz per GPU, x_i = 0.25 i / n",
                       "rows": n_local * world, "nnz": int(nnz_total), "parallelism": f"row-blocks x{world}",
                       "exchange": {"none": "none",
                                    "p2p_fused": f"inside the product kernel: its seam workgroups load {peer.n_halo if peer else 0} "
                                                 "halo entries over xGMI from the neighbours' shared x slices "
                                                 "(one launch per step, no collective)",
                                    "p2p": f"per step: pull of {peer.n_halo if peer else 0} halo entries over xGMI from the "
                                           "neighbours' shared x slices (one kernel, no collective)",
                                    "all_gather": "per step: RCCL all_gather(x)"}[exchange],
                       "rows_wrong_vs_oracle_all_ranks": rows_wrong,
                       "matrix_copies_rotated": copies, "launch": launch_mode, "untimed_preroll_replays": preroll, "design_point": design,
                       "grid": info.grid, "lds_bytes": info.lds_bytes, "tune": tune_info},
            "hbm_gbs_algorithmic": round(achieved * world, 1),
            "hbm_pct_of_peak": round(100.0 * achieved / HBM_PEAK_GBS, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_spmv_" + design["variant"], "algorithmic_bytes_per_launch": alg_bytes,
                         "launch_usec": round(launch_us, 3)},
            "warm_cache": {"usec_graph": round(warm_graph, 3) if warm_graph else None,
                           "gflops_graph": round(2.0 * nnz_local / warm_graph * 1e-3, 2) if warm_graph else None,
                           "usec_eager_median": round(warm_med, 3), "usec_eager_min": round(warm_min, 3),
                           "note": "ONE copy of the matrix replayed (what a solver iteration sees; the Infinity Cache "
                                   "holds it, not an HBM figure): usec_graph is measured like `value` (graph of 200 "
                                   "launches), usec_eager_* are single launches between their own event pairs"},
        }
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(rp, ci, va, x_host, y_gpu, args.cpu_seconds)
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)
    if use_dist:
        host_barrier()
        if peer is not None:
            for ptr in peer.peers.values():
                p2p.close_peer(ptr)
            peer.peers = {}
            host_barrier()                                       # owners free only after every peer has unmapped
            peer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
