#!/usr/bin/env python3
"""Headline benchmark: fp64 CSR SpMV on the cant-like matrix (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--workload W] [--solver S]

A "step" is one pass of the hot path over data resident in HBM:

* default (``--workload cant``): y = A x over the whole matrix.  N > 1 is WEAK scaling: every rank owns one
  cant-sized row block of a block-banded global matrix and the slice of x that goes with it; a step is ONE launch
  per rank -- the seam workgroups of the product kernel load the halo entries they need straight from the
  neighbours' shared x slices over xGMI (cask_hip_csr_set_halo_sources), no collective on the data path.
* ``--workload webbase-1M`` (BASELINE configs[3]), ``G3_circuit``, ``atmosmodd``: ONE global matrix, rows dealt to the
  ranks in nnz-balanced contiguous blocks (STRONG scaling).  The exchange is chosen per matrix from the halo
  fraction: where a block references most of x (webbase-like: 30 % uniformly random columns) a step is an RCCL
  all-gather of x + the local product; banded / stencil matrices read their halos in-kernel.
  ``CASK_BENCH_EXCHANGE=all_gather|p2p`` forces one.
* ``--solver cg|bicg`` (configs[2] and [4]: ``--workload G3_circuit --solver cg``, ``--workload atmosmodd --solver
  bicg``): a step is one solver pass run by the engine (cask_hip_solve_device: fused product + dot, update kernels,
  device-resident scalars); row-sharded for N > 1 with the dot products all-reduced over RCCL.  The timed solve
  runs exactly K passes (tol = 0); a separate solve to tol 1e-5 is checked against the oracle.

To make the SpMV number an HBM number and not an Infinity-Cache number the steps rotate through enough device
copies of the matrix to exceed 2x the 256 MiB cache ("cold"); the cache-warm rate of one copy is reported next to
it.  The K timed steps are captured once into a HIP graph where the step has no collective.

ONE clock: `value`, `ms_per_step` and `roofline.*` all come from the HIP-event time of the K timed steps (events on
the launch stream, between the two barrier + synchronize brackets; the MAX over ranks).  The host wall time of the
same region is reported as `host_wall_ms_per_step`.

Prints ONE JSON line on rank 0 with
  roofline      -- algorithmic bytes per launch / mean launch time against the 8 TB/s HBM peak
  cpu_baseline  -- MKL's dcsrgemv (the CPU library the reference calls) on pinned host threads, and the 1-core
                   C restatement (oracle) as the secondary figure; both on the same matrix, bounded to seconds.
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
EXIT_NOT_DISTINCT = 4             # N > 1 ranks on fewer than N devices and no CASK_BENCH_SHARE_DEVICE: nothing was timed
EXIT_OTHERS_FAILED = 3            # the headline line was printed, an appended workload failed or hung (never a restart in-process)
HALO_FRACTION_FOR_ALLGATHER = 0.10   # a block that references more than this share of x gets the all-gather:
#   per-element remote loads pay a fabric packet per 8 useful bytes, the all-gather moves whole slices (webbase-like
#   blocks reference 16 % of x scattered over every peer; stencil / banded blocks 0.1-2 % from their neighbours)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cant", choices=["cant", "cant3", "G3_circuit", "webbase-1M", "webbase2", "atmosmodd"])
    ap.add_argument("--solver", default=None, choices=["cg", "bicg"], help="time solver passes instead of products")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "sequence", "eager"],
                    help="graph: the K steps as one HIP graph; sequence: K launches from one C call (no graph start-up "
                         "inside a short timed region); eager: per-step calls; auto: sequence below 200 steps, else graph")
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of K steps each, back to back behind one pre-roll; ms_per_step is the MEDIAN window "
                         "/ K (0 = auto: at least 31, more while K*windows < 2000 steps, at most 101)")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="sustained untimed load in front of the timed windows (the step time settles ~150 ms after the GPU "
                         "leaves idle, the shader clock ~500 ms: tools/drift_probe.py)")
    ap.add_argument("--no-tune", action="store_true", help="skip the measured DSE, use the AUTO design point")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--copies", type=int, default=0, help="matrix copies to rotate through (0 = auto)")
    ap.add_argument("--variant", default=None, help="force a design point: vector|merge|merge_wave|scan")
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--wg", type=int, default=0)
    ap.add_argument("--no-others", action="store_true",
                    help="headline workload only (the default cant run otherwise appends the other BASELINE configs "
                         "under config.other_workloads)")
    ap.add_argument("--strict-exit", action="store_true",
                    help="exit with status 3 (after printing the headline line) when an appended workload failed or hung")
    ap.add_argument("--other-steps", type=int, default=200, help="timed steps of each appended workload")
    ap.add_argument("--other-seconds", type=float, default=200.0,
                    help="watchdog for the appended workloads as a whole: past it the headline line is printed without them")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------- CPU baselines
def load_mkl():
    for cand in (os.environ.get("MKLROOT", "/nonexistent") + "/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.1",
                 "/opt/conda/lib/libmkl_rt.so.2"):
        if os.path.exists(cand):
            try:
                return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
    return None


def cpu_baseline(rp, ci, va, x, y_gpu, seconds, quick=False):
    """MKL mkl_cspblas_dcsrgemv on pinned threads (primary: the reference's CPU library) and the 1-core oracle.
    `quick` (the appended workloads of the default line: a bounded ~0.5 s each): ONE team size (16 threads, the size the
    headline's sweep picks on this host class), the full-matrix routine only, a short sample of the 1-core port."""
    import oracle
    n = rp.size - 1
    nnz = int(ci.size)
    y = oracle.csr_spmv(rp, ci, va, x)                  # warm-up + the checker
    bad, _ = oracle.mismatches(y_gpu, y) if y_gpu is not None else (None, None)
    t0 = time.perf_counter()
    calls = 0
    budget = min(seconds, 5.0) if not quick else min(seconds, 0.15)
    while True:
        oracle.csr_spmv(rp, ci, va, x)
        calls += 1
        el = time.perf_counter() - t0
        if el >= budget or calls >= 5000:
            break
    port = {"value": round(2.0 * nnz * calls / el / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"{calls} sequential CSR SpMVs of the same {n}x{x.size} matrix in {el:.1f} s (oracle/cask_oracle.c)"}
    out = dict(port)
    out["parity_gpu_vs_cpu_mismatches"] = bad
    mkl = load_mkl()
    if mkl is not None and x.size == n:
        try:
            mkl.MKL_Get_Max_Threads.restype = ctypes.c_int
            max_threads = int(mkl.MKL_Get_Max_Threads())
            tr, nn = ctypes.c_char(b"N"), ctypes.c_int(n)
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
            ym = np.zeros(n)
            # the two products the reference's CPU code calls: mkl_cspblas_dcsrgemv on the full matrix
            # (fpgaNaiveCpuCode.cpp:33) and, for a symmetric matrix, mkl_dcsrsymv('l') on the stored triangle
            # (SparseLinearSolvers.hpp:189); both count the full matrix's 2*nnz flops
            routines = {"mkl_cspblas_dcsrgemv": (mkl.mkl_cspblas_dcsrgemv,
                                                 (ctypes.byref(tr), ctypes.byref(nn), p(va), p(rp), p(ci), p(x), p(ym)))}
            if not quick and is_symmetric(rp, ci, va):
                lo = ctypes.c_char(b"l")
                lrp, lci, lva = lower_triangle_1based(rp, ci, va)
                ys = np.zeros(n)
                routines["mkl_dcsrsymv('l')"] = (mkl.mkl_dcsrsymv, (ctypes.byref(lo), ctypes.byref(nn), p(lva), p(lrp),
                                                                   p(lci), p(x), p(ys)))
            # the host is shared and a 62 K-row product does not feed 128 threads: time a few team sizes for
            # a slice of the budget each and report the best one (threads pinned: OMP_PROC_BIND/OMP_PLACES below)
            counts = sorted({t for t in (8, 16, 32, 64, 128, 256) if 0 < t <= max_threads} | {min(max_threads, 16)})
            if quick:
                counts = [min(max_threads, 16)]
            by_routine, best = {}, None
            for name, (fn, args) in routines.items():
                by_threads = by_routine.setdefault(name, {})
                for t in counts:
                    mkl.MKL_Set_Num_Threads(ctypes.c_int(t))
                    for _ in range(3):
                        fn(*args)
                    t0 = time.perf_counter()
                    calls = 0
                    while True:
                        fn(*args)
                        calls += 1
                        el = time.perf_counter() - t0
                        if el >= seconds / (len(counts) * len(routines)) or calls >= 20000:
                            break
                    rate = 2.0 * nnz * calls / el / 1e9
                    by_threads[str(t)] = round(rate, 3)
                    if best is None or rate > best[0]:
                        best = (rate, t, calls, el, name)
            mbad, _ = oracle.mismatches(ym, y)
            if len(routines) > 1:
                mbad += oracle.mismatches(ys, y)[0]
            out = {"value": round(best[0], 4), "unit": "GFLOP/s", "cores": best[1], "kind": "mkl",
                   "routine": f"{best[4]} (oneMKL, GNU threading layer, OMP_PROC_BIND=close OMP_PLACES=cores)",
                   "host_cores": os.cpu_count(), "sample": f"{best[2]} calls with {best[1]} threads in {best[3]:.1f} s",
                   "gflops_by_routine_and_threads": by_routine, "mismatches_vs_oracle": mbad,
                   "parity_gpu_vs_cpu_mismatches": bad, "port": port}
        except Exception as e:  # pragma: no cover - diagnostic only
            out["mkl_error"] = repr(e)
    return out


def is_symmetric(rp, ci, va):
    """A == A^T, entry for entry (square matrices; one counting-sort transpose on the host)."""
    from cask_amd import dist as cdist
    n = rp.size - 1
    trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
    return bool(np.array_equal(trp, rp) and np.array_equal(tci, ci) and np.array_equal(tva, va))


def lower_triangle_1based(rp, ci, va):
    """The stored triangle as the reference hands it to mkl_dcsrsymv: lower + diagonal, 1-based (SymCsrMatrix,
    SparseLinearSolvers.hpp:171-189)."""
    n = rp.size - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    keep = ci <= rows
    lrp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows[keep], minlength=n), out=lrp[1:])
    return lrp + 1, (ci[keep] + 1).astype(np.int32), np.ascontiguousarray(va[keep])


def mkl_solver_passes(mkl, kind, rp, ci, va, b, passes, threads, sym=None):
    """`passes` passes of the reference's CG recurrence (pcg, SparseLinearSolvers.hpp:200-232: product, ddot, daxpy,
    daxpby) -- or of BiCG with the transposed product -- on MKL calls.  The product is mkl_cspblas_dcsrgemv on the full
    matrix, or, with `sym` = lower_triangle_1based(...), mkl_dcsrsymv('l') on the stored triangle exactly as the
    reference's CPU path calls it (:189,206).  Seconds."""
    n = rp.size - 1
    mkl.MKL_Set_Num_Threads(ctypes.c_int(threads))
    mkl.cblas_ddot.restype = ctypes.c_double
    p_ = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    nn, trN, trT, lo = ctypes.c_int(n), ctypes.c_char(b"N"), ctypes.c_char(b"T"), ctypes.c_char(b"l")

    def gemv(tr, x, y):
        if sym is not None:
            mkl.mkl_dcsrsymv(ctypes.byref(lo), ctypes.byref(nn), p_(sym[2]), p_(sym[0]), p_(sym[1]), p_(x), p_(y))
        else:
            mkl.mkl_cspblas_dcsrgemv(ctypes.byref(tr), ctypes.byref(nn), p_(va), p_(rp), p_(ci), p_(x), p_(y))

    dot = lambda x, y: mkl.cblas_ddot(n, p_(x), 1, p_(y), 1)  # noqa: E731
    axpy = lambda a, x, y: mkl.cblas_daxpy(n, ctypes.c_double(a), p_(x), 1, p_(y), 1)  # noqa: E731
    axpby = lambda a, x, bb, y: mkl.cblas_daxpby(n, ctypes.c_double(a), p_(x), 1, ctypes.c_double(bb), p_(y), 1)  # noqa: E731
    x, r = np.zeros(n), b.copy()
    p, q = r.copy(), np.zeros(n)
    if kind == "bicg":
        rt, pt, qt = r.copy(), r.copy(), np.zeros(n)
    rho = dot(r, r)
    t0 = time.perf_counter()
    for _ in range(passes):
        gemv(trN, p, q)
        if kind == "cg":
            alpha = rho / max(dot(p, q), 1e-300)
            axpy(alpha, p, x)
            axpy(-alpha, q, r)
            rho_new = dot(r, r)
            axpby(1.0, r, rho_new / max(rho, 1e-300), p)
        else:
            gemv(trT, pt, qt)
            alpha = rho / max(dot(pt, q), 1e-300)
            axpy(alpha, p, x)
            axpy(-alpha, q, r)
            axpy(-alpha, qt, rt)
            dot(r, r)
            rho_new = dot(rt, r)
            beta = rho_new / (rho if rho != 0 else 1e-300)
            axpby(1.0, r, beta, p)
            axpby(1.0, rt, beta, pt)
        rho = rho_new if np.isfinite(rho_new) and rho_new != 0 else 1.0
    return time.perf_counter() - t0


def cpu_baseline_solver(kind, rp, ci, va, b, seconds, quick=False):
    """`quick`: the appended workloads' bounded sample -- 16 pinned threads, the full-matrix product only, no 1-core port.
    MKL (the reference's CPU path: pcg on mkl_dcsrsymv + cblas, SparseLinearSolvers.hpp:162-239) on pinned threads,
    and the oracle's CG / BiCG (1 core) as the secondary figure; a bounded number of passes each; GFLOP/s on the same
    flop count as `value` (the full matrix's 2*nnz per product, whichever routine ran).  CG is timed with both products
    the reference's CPU code knows -- mkl_dcsrsymv('l') on the stored triangle (what pcg calls) and
    mkl_cspblas_dcsrgemv on the full matrix -- and the faster one is `value`."""
    port = cpu_baseline_solver_port(kind, rp, ci, va, b, min(seconds, 4.0)) if not quick else None
    mkl = load_mkl()
    if mkl is None:
        return port
    try:
        n, nnz = rp.size - 1, int(ci.size)
        flops = (2 * nnz + 12 * n) if kind == "cg" else (4 * nnz + 20 * n)
        mkl.MKL_Get_Max_Threads.restype = ctypes.c_int
        max_threads = int(mkl.MKL_Get_Max_Threads())
        routines = {"mkl_cspblas_dcsrgemv": None}
        if kind == "cg" and not quick:
            routines["mkl_dcsrsymv('l')"] = lower_triangle_1based(rp, ci, va)
        counts = sorted({t for t in (8, 16, 32, 64, 128, 256) if t <= max_threads} | {min(16, max_threads)})
        if quick:
            counts = [min(16, max_threads)]
        by_routine, best = {}, None
        for name, sym in routines.items():
            by_threads = by_routine.setdefault(name, {})
            for t in counts:
                mkl_solver_passes(mkl, kind, rp, ci, va, b, 2 if quick else 3, t, sym)   # warm-up
                passes = 4 if quick else 20
                el = mkl_solver_passes(mkl, kind, rp, ci, va, b, passes, t, sym)
                share = seconds / len(routines)
                if el < share / 8:
                    passes = int(min(2000, passes * (share / 4) / max(el, 1e-4)))
                    el = mkl_solver_passes(mkl, kind, rp, ci, va, b, passes, t, sym)
                rate = flops * passes / el / 1e9
                by_threads[str(t)] = round(rate, 3)
                if best is None or rate > best[0]:
                    best = (rate, t, passes, el, name)
        return {"value": round(best[0], 4), "unit": "GFLOP/s", "cores": best[1], "kind": "mkl",
                "routine": f"{kind} passes on {best[4]} + cblas_ddot/daxpy/daxpby (oneMKL, GNU threading layer, pinned)",
                "host_cores": os.cpu_count(), "sample": f"{best[2]} passes with {best[1]} threads in {best[3]:.1f} s",
                "gflops_by_routine_and_threads": by_routine, "port": port}
    except Exception as e:  # pragma: no cover - diagnostic only
        port = port or {}
        port["mkl_error"] = repr(e)
        return port


def cpu_baseline_solver_port(kind, rp, ci, va, b, seconds):
    """The oracle's CG / BiCG (1 core) for a bounded number of passes."""
    import oracle
    n, nnz = rp.size - 1, int(ci.size)
    passes = 10
    fn = oracle.cg_full if kind == "cg" else oracle.bicg
    t0 = time.perf_counter()
    fn(rp, ci, va, b, maxiters=passes, tol=0.0)
    el = time.perf_counter() - t0
    if el < seconds / 4:                                    # scale the sample up to a few seconds
        passes = int(min(200, passes * seconds / 2 / max(el, 1e-3)))
        t0 = time.perf_counter()
        fn(rp, ci, va, b, maxiters=passes, tol=0.0)
        el = time.perf_counter() - t0
    flops = (2 * nnz + 12 * n) if kind == "cg" else (4 * nnz + 20 * n)
    return {"value": round(flops * passes / el / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"{passes} {kind} passes of the same {n}-row system in {el:.1f} s (oracle/cask_oracle.c)"}


# ------------------------------------------------------------------------------------------------- main
def self_launch(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this process -- which has not imported torch or
    touched the GPU -- starts the N ranks as CHILD processes (never an exec), one per GPU, with the rendezvous
    variables torch.distributed.run would set, relays rank 0's JSON line (rank 0 inherits stdout) and exits with the
    children's status."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = list(procs)
    try:
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:                       # one rank failed: the others would wait in a collective
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for q in pending:
            q.kill()
        rc = 130
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         "(or leave WORLD_SIZE unset and bench.py starts them itself)")
    # CASK_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, all_gather, all_reduce) with a single
    # rank: the RCCL path on a 1-GPU box.
    use_dist = world > 1 or bool(os.environ.get("CASK_BENCH_FORCE_DIST"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL and the shared x slices need on this pool
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

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

    class Ctx:
        pass
    cx = Ctx()
    cx.args, cx.rank, cx.world, cx.dev, cx.use_dist, cx.backend, cx.ctl_dev = args, rank, world, dev, use_dist, backend, ctl_dev

    def all_reduce_scalar(v, op):
        if not use_dist:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=op)
        return float(t[0])

    def host_barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
    cx.all_reduce_scalar, cx.host_barrier = all_reduce_scalar, host_barrier
    t_start = time.perf_counter()
    cx.rccl = rccl_census(cx, dist, capi, local_rank) if use_dist else None

    def phase(name, since):
        """Per-phase wall budget on stderr (rank 0): a multi-GPU run that comes close to the harness's limit says where."""
        if rank == 0 and (world > 1 or os.environ.get("CASK_BENCH_PHASES")):
            now = time.perf_counter()
            print(f"[bench] {name}: {now - since:.1f} s (run so far {now - t_start:.1f} s)", file=sys.stderr, flush=True)
    cx.phase = phase

    def run_one(a):
        c = Ctx()
        c.__dict__.update(cx.__dict__)
        c.args = a
        if a.solver:
            return run_solver(c)
        return run_spmv(c, weak=a.workload in ("cant", "cant3"))

    t_head = time.perf_counter()
    rec = run_one(args)
    phase("headline (everything above, + post-run check, warm-cache figures, CPU baseline)", t_head)
    # ---- the other BASELINE configs next to the headline (default cant run only) ------------------------------
    if args.workload == "cant" and not args.solver and not args.no_others:
        import threading
        state = {"done": False}
        others = []
        if rank == 0:
            rec["config"]["other_workloads"] = others

        def give_up():                                       # a hung collective in an appended workload must not
            if state["done"]:                                # cost the headline: print what there is and leave
                return
            if rank == 0:
                others.append({"error": f"watchdog: appended workloads exceeded {args.other_seconds:.0f} s"})
                rec["appended_workloads_failed"] = True
                print(json.dumps(rec), flush=True)
            os._exit(others_exit_status(args))               # the headline went out
        dog = threading.Timer(args.other_seconds, give_up)
        dog.daemon = True
        dog.start()
        for spec in other_workload_specs(world):
            a = argparse.Namespace(**vars(args))
            a.workload, a.solver = spec["workload"], spec.get("solver")
            a.steps, a.warmup = args.other_steps, max(2, args.other_steps // 10)
            # the MKL column next to every appended workload too (N = 1): a bounded ~0.5 s sample each,
            # 16 pinned threads, inside the watchdog's budget
            a.no_cpu_baseline, a.launch, a.copies = args.no_cpu_baseline or world > 1, "auto", 0
            a.cpu_seconds, a.cpu_quick = 0.5, True
            a.variant, a.lanes, a.tile, a.items, a.wg, a.far = None, 0, 0, 0, 0, 0
            t0 = time.perf_counter()
            try:
                sub = run_one(a)
                phase(f"appended workload {spec['workload']}" + (f" --solver {spec['solver']}" if spec.get("solver") else ""), t0)
                if rank == 0:
                    others.append(summarise_other(spec, sub, time.perf_counter() - t0))
            except Exception as e:  # noqa: BLE001 - reported in the line, the headline stands
                others.append({"config": spec["config"], "workload": spec["workload"], "error": repr(e)})
                print(f"[bench] rank {rank}: appended workload {spec['workload']} failed: {e!r}", file=sys.stderr, flush=True)
                if use_dist:
                    # this rank has left a collective sequence its peers are still in: it must not enter another one.
                    # Its line (rank 0) goes out now; the peers' watchdogs end them with the headline intact.
                    state["done"] = True
                    if rank == 0:
                        rec["appended_workloads_failed"] = True
                        print(json.dumps(rec), flush=True)
                    os._exit(others_exit_status(args))
        state["done"] = True
        dog.cancel()
        failed = any("error" in o for o in others)
    else:
        failed = False
    if rank == 0:
        if failed:
            rec["appended_workloads_failed"] = True
        print(json.dumps(rec), flush=True)
    if use_dist:
        host_barrier()
        dist.destroy_process_group()
    if failed and others_exit_status(args):
        sys.exit(others_exit_status(args))


def rccl_census(cx, dist, capi, local_rank):
    """What the collectives layer really saw, gathered before anything is timed: the backend, the
    world size of the process group, ncclCommCount / the device of the ENGINE's own communicator (the one the sharded
    solvers issue their all-reduces on), and the PCI bus id of every rank's GPU.  A run in which two ranks of one host
    open the SAME device index is refused -- non-zero exit, no timing -- unless CASK_BENCH_SHARE_DEVICE says it is a dry
    run on a shared device.  The refusal covers index collisions only (r6, ADVICE r5: the wording used to promise more):
    PCI bus ids that collide while the indices differ -- a virtualised bus -- are REPORTED (`distinct_devices` < world),
    not refused, so that a real node is never turned away by a naming quirk.  The line's config.rccl makes "RCCL saw N
    ranks on N devices" checkable from the JSON alone."""
    rank, world = cx.rank, cx.world
    try:
        pci = capi.device_pci_bus_id(local_rank)
    except Exception as e:  # noqa: BLE001
        pci = f"unknown ({e!r})"
    mine = {"rank": rank, "local_rank": local_rank, "pci_bus_id": pci, "host": os.uname().nodename}
    comm = None
    if cx.backend == "nccl" and not os.environ.get("CASK_NO_NATIVE_RCCL"):
        uid = [None]
        if rank == 0:
            try:
                uid[0] = capi.NativeComm.unique_id()
            except Exception as e:  # noqa: BLE001 - agreed on below
                mine["engine_comm_error"] = repr(e)
        dist.broadcast_object_list(uid, src=0)
        # ncclCommInitRank is a collective: a rank that cannot take part (its library did not load) would leave the others
        # blocked inside RCCL, in front of the all_gather_object that reports the error -- so the ranks first AGREE that
        # every one of them can
        ready = 0
        if uid[0] is not None:
            try:
                capi.NativeComm._lib()
                ready = 1
            except Exception as e:  # noqa: BLE001
                mine["engine_comm_error"] = repr(e)
        flags = [None] * world
        dist.all_gather_object(flags, ready)
        if all(flags):
            try:
                comm = capi.NativeComm(uid[0], rank, world)
                mine.update(comm.info())
            except Exception as e:  # noqa: BLE001
                mine["engine_comm_error"] = repr(e)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if comm is not None:
        comm.close()
    devices = [f"{e['host']}:{e['pci_bus_id']}" for e in everyone]
    # ranks that open the SAME device index on the same host share a device for certain; PCI ids that collide while the
    # indices differ (a virtualised bus) are reported, not refused -- a real node must never be turned away by a naming quirk
    opened = [f"{e['host']}:index{e['local_rank']}" for e in everyone]
    counts = sorted({e.get("comm_nranks") for e in everyone if e.get("comm_nranks") is not None})
    info = {"backend": cx.backend + (" (RCCL)" if cx.backend == "nccl" else ""), "world_size_seen": dist.get_world_size(),
            "comm_nranks": counts[0] if len(counts) == 1 else (counts or None),
            "comm_devices": [e.get("device") for e in everyone] if counts else None,
            "devices": devices, "distinct_devices": len(set(devices)), "distinct_device_indices": len(set(opened)),
            "shared_device_dry_run": bool(os.environ.get("CASK_BENCH_SHARE_DEVICE")),
            "errors": [f"rank {e['rank']}: {e['engine_comm_error']}" for e in everyone if e.get("engine_comm_error")] or None}
    if info["distinct_device_indices"] < world and not info["shared_device_dry_run"]:
        if rank == 0:
            print(f"[bench] {world} ranks on {info['distinct_device_indices']} distinct devices ({devices}): refusing to time a "
                  "multi-GPU run that is not one (set CASK_BENCH_SHARE_DEVICE=1 for a dry run on a shared device)",
                  file=sys.stderr, flush=True)
        # an orderly end on EVERY rank (all of them hold the same gathered `info`): a rank that simply exits while a peer's
        # gloo threads still hold its sockets can make that peer abort (SIGABRT instead of status 4: seen once in r6)
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        raise SystemExit(EXIT_NOT_DISTINCT)
    return info


def xgmi_fields(exchange, world, step_us, stride=0, halo_by_owner=None):
    """Bytes a step moves over xGMI per GPU and per link, and the rate that is at the measured step time (SURVEY 8d/8e:
    the all-gather receives 8 n_cols (G - 1) / G bytes per GPU -- one slice over each of the G - 1 direct links; the
    in-kernel halo / pull only the referenced entries, from the neighbours that own them)."""
    if world <= 1 or exchange == "none":
        return None
    if exchange in ("all_gather", "push"):
        per_link = 8 * int(stride)
        received = per_link * (world - 1)
        what = f"{exchange}: a slice of {stride} doubles (padded stride) from each of {world - 1} peers, one direct link each"
    else:
        by = [int(v) for v in (halo_by_owner or [])]
        per_link = 8 * max(by) if by else 0
        received = 8 * sum(by)
        what = f"{exchange}: {sum(by)} halo entries from {sum(1 for v in by if v)} peers (rank 0's block)"
    return {"what": what, "bytes_received_per_step_per_gpu": received, "bytes_per_link_max": per_link,
            "gbs_per_link_at_step_time": round(per_link / (step_us * 1e-6) / 1e9, 2) if step_us > 0 else None,
            "link_peak_gbs": 153.0}


def others_exit_status(args):
    """Exit status once the headline line is out and an appended workload failed or hung.  The line then carries
    `appended_workloads_failed: true` and the error entries; with --strict-exit (or CASK_BENCH_STRICT_EXIT=1) the process
    also ends with status 3 so that a harness which only looks at exit codes sees it.  Not the default: the driver's
    contract is ONE JSON line from a run it can use, and the headline measurement in that line is complete and valid."""
    return EXIT_OTHERS_FAILED if (args.strict_exit or os.environ.get("CASK_BENCH_STRICT_EXIT")) else 0


def other_workload_specs(world):
    """What the default run times after the headline: at N = 1 the other BASELINE configs in their 1-GPU form, at N > 1
    the strong-scaling configs[3] and [4] (+ CG on G3_circuit, configs[2] sharded)."""
    if world == 1:
        return [{"config": "configs[1] second look-alike", "workload": "cant3"},
                {"config": "configs[3] on 1 GPU", "workload": "webbase-1M"},
                {"config": "configs[3] second look-alike (hub columns, site-block locality) on 1 GPU", "workload": "webbase2"},
                {"config": "configs[2]", "workload": "G3_circuit", "solver": "cg"},
                {"config": "configs[4] on 1 GPU", "workload": "atmosmodd", "solver": "bicg"}]
    return [{"config": "configs[3]", "workload": "webbase-1M"},
            {"config": "configs[4]", "workload": "atmosmodd", "solver": "bicg"},
            {"config": "configs[2] row-sharded", "workload": "G3_circuit", "solver": "cg"}]


def summarise_other(spec, sub, seconds):
    c, roof = sub["config"], sub["roofline"]
    out = {"config": spec["config"], "workload": c["workload"], "metric": sub["metric"], "value": sub["value"],
           "unit": sub["unit"], "usec": roof["launch_usec"], "steps": sub["steps"], "scaling": sub["scaling"],
           "frac": roof["frac"], "hbm_gbs_algorithmic_per_gpu": roof["achieved"],
           "algorithmic_bytes_per_step_per_gpu": roof["algorithmic_bytes_per_launch"],
           "traffic": roof.get("traffic"), "traffic_source": roof.get("traffic_source"), "exchange": c.get("exchange"), "design_point": c.get("design_point"),
           "windows": sub.get("windows"), "usec_p10": round(sub.get("ms_per_step_p10", 0) * 1e3, 3),
           "usec_p90": round(sub.get("ms_per_step_p90", 0) * 1e3, 3), "seconds_in_bench": round(seconds, 1)}
    if roof.get("working_set_note"):
        out["working_set_note"] = roof["working_set_note"]
    cb = sub.get("cpu_baseline")
    if cb:                                                   # the MKL column of this workload (bounded sample, 16 threads)
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "routine", "sample", "host_cores",
                                                       "mismatches_vs_oracle", "parity_gpu_vs_cpu_mismatches", "mkl_error")
                               if cb.get(k) is not None}
    if "solve_check" in c:
        out["solve_check"] = c["solve_check"]
        out["collectives"] = c.get("collectives")
        out["exchange_selfcheck"] = c.get("exchange_selfcheck")
    else:
        out["rows_wrong"] = c.get("rows_wrong_vs_oracle_all_ranks")
        out["exchange_selfcheck"] = c.get("exchange_selfcheck")
        out["matrix_copies_rotated"] = c.get("matrix_copies_rotated")
        out["launch"] = c.get("launch")
    return out


def n_windows(args):
    """Windows of K timed steps (the same on every rank: a function of the arguments alone)."""
    if args.windows > 0:
        return args.windows
    r = 31
    while r * max(args.steps, 1) < 2000 and r < 101:
        r += 10
    return r


def gpu_clocks(dev_index=0):
    """Current shader / memory clock of the device in MHz from sysfs (pp_dpm_sclk / pp_dpm_mclk: the starred level), so
    that a slow box explains itself in the line.  None where the files cannot be read."""
    out = {}
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev_index)
        bus = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0"
        roots = [Path("/sys/bus/pci/devices") / bus]
    except Exception:  # noqa: BLE001 - reporting only
        roots = []
    if not roots or not (roots[0] / "pp_dpm_sclk").exists():
        roots = sorted(Path("/sys/class/drm").glob("card[0-9]*/device"))
    for root in roots:
        got = {}
        for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
            try:
                for line in (root / name).read_text().splitlines():
                    if "*" in line:
                        got[key] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            except Exception:  # noqa: BLE001
                pass
        if got:
            out = got
            break
    return out or None


def timed_windows(cx, run_steps, lead_in=None, windows=31, native=None):
    """R windows of exactly K timed steps each, back to back on the launch stream between the two barrier + synchronize
    brackets: R + 1 HIP events, window r = event r -> event r + 1.  ``lead_in`` (an untimed run of the same K steps) is
    queued in front of the first event so that the timed work is already submitted when that event fires.  Returns a
    dict: per-window device ms of THIS rank, their median / min / p10 / p90 / max, the MAX over ranks of the median
    (`dev_ms`: the figure ms_per_step is derived from -- ONE 20-step window is a 0.2 ms sample and moves by 10 % from box
    to box and run to run; the median of >= 31 of them does not), the whole region's device time and host wall time."""
    import torch
    import torch.distributed as dist
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
    cx.host_barrier()
    torch.cuda.synchronize()
    if lead_in is not None:
        lead_in()
    t0 = time.perf_counter()
    if native is not None:
        # the engine's own windows: K launches + one timing event per window from ONE C call, events without the
        # system-scope fence a torch event carries (cask_hip_spmv_windows_device); returns when the last has completed
        ms = [u * 1e-3 for u in native(windows)]
        t_submit = time.perf_counter() - t0
        clocks_busy = gpu_clocks(cx.dev.index or 0)
        cx.host_barrier()
        wall = time.perf_counter() - t0
        total = float(sum(ms))
    else:
        ev[0].record()
        for r in range(windows):
            run_steps()
            ev[r + 1].record()
        t_submit = time.perf_counter() - t0
        clocks_busy = gpu_clocks(cx.dev.index or 0)        # read while the windows run (the host is ahead of the stream)
        cx.host_barrier()
        wall = time.perf_counter() - t0
        ms = [ev[r].elapsed_time(ev[r + 1]) for r in range(windows)]
        total = ev[0].elapsed_time(ev[windows])
    srt = sorted(ms)
    q = lambda f: srt[min(windows - 1, max(0, int(round(f * (windows - 1)))))]   # noqa: E731
    med = float(np.median(ms))
    out = {"windows": windows, "ms": ms, "median": med, "min": srt[0], "p10": q(0.10), "p90": q(0.90), "max": srt[-1],
           "mean": total / windows, "first": ms[0], "host_submit_s": t_submit, "clocks_during": clocks_busy}
    if cx.use_dist:
        out["dev_ms"] = cx.all_reduce_scalar(med, dist.ReduceOp.MAX)
        out["mean_max"] = cx.all_reduce_scalar(out["mean"], dist.ReduceOp.MAX)
        out["wall"] = cx.all_reduce_scalar(wall, dist.ReduceOp.MAX)
    else:
        out["dev_ms"], out["mean_max"], out["wall"] = med, out["mean"], wall
    return out


def window_fields(tw, steps):
    """The spread of the timed windows for the JSON line (ms per step, like ms_per_step)."""
    k = max(steps, 1)
    r6 = lambda v: round(v / k, 6)                          # noqa: E731
    return {"windows": tw["windows"], "ms_per_step_min": r6(tw["min"]), "ms_per_step_p10": r6(tw["p10"]),
            "ms_per_step_p90": r6(tw["p90"]), "ms_per_step_max": r6(tw["max"]), "ms_per_step_mean": r6(tw["mean_max"]),
            "ms_per_step_first_window": r6(tw["first"]),
            "window_spread_pct": round(100.0 * (tw["p90"] - tw["p10"]) / max(tw["median"], 1e-12), 2)}


def traffic_record(workload, design=None):
    """Counter-measured HBM bytes per launch of an earlier profiled run of the same workload (not measured here) --
    ONLY from the file keyed by THIS design point (tools/pmc_point.sh, tools/dse_evidence.py).  When the timed point was
    never profiled the line says `traffic: null` and names the nearest file apart (r6, VERDICT r5 item 7: round 5's
    webbase-1M line reported the bytes of `scan w512 i4` for a run that timed `scan w256 i8`).  A solver workload has
    no design point in its key: its one file is exact."""
    def load(name):
        tfile = REPO / "profiles" / name
        if not tfile.exists():
            return None
        try:
            return json.loads(tfile.read_text())
        except Exception:
            return None
    how = "rocprofv3 PMC passes of round {tag}, FETCH_SIZE x calibration + WRITE_SIZE; not measured in this run"
    if design is None:
        t = load(f"traffic_{workload}.json")
        if t is None:
            return None, None
        return t.get("hbm_bytes_per_launch"), f"profiles/traffic_{workload}.json (" + how.format(tag=t.get("tag", "?")) + ")"
    label = f'{design["variant"]}_w{design["wg_size"]}_i{design["items_per_thread"]}_t{design["tile_width"]}' \
            f'_l{design["lanes_per_row"]}'.replace("-", "m")
    t = load(f"traffic_{workload}_{label}.json")
    if t is not None:
        return t.get("hbm_bytes_per_launch"), f"profiles/traffic_{workload}_{label}.json (this design point; " + how.format(tag=t.get("tag", "?")) + ")"
    near = load(f"traffic_{workload}.json")
    if near is None:
        return None, f"no counter file for design point {label}"
    return None, (f"no counter file for the timed design point {label}; nearest: profiles/traffic_{workload}.json = "
                  f"{near.get('hbm_bytes_per_launch')} bytes per launch at {near.get('design_point', 'that round AUTO point')} "
                  f"(round {near.get('tag', '?')}) -- another point, so not reported as this run's traffic")


def run_spmv(cx, weak):
    import torch
    import torch.distributed as dist
    from cask_amd import capi, synth
    from cask_amd import dist as cdist
    args, rank, world, dev, use_dist = cx.args, cx.rank, cx.world, cx.dev, cx.use_dist
    from cask_amd import selfcheck
    selfchecks = {}
    phase = cx.phase

    # ---- workload -----------------------------------------------------------
    t_gen = time.perf_counter()
    if weak:
        gen = synth.cant_like_shard if args.workload == "cant" else synth.cant3_like_shard
        if world == 1 and os.environ.get("CASK_MATRIX_DIR") and (Path(os.environ["CASK_MATRIX_DIR"]) / "cant.mtx").exists():
            n_local, rp, ci, va, source = synth.load_or_make("cant")
            n_global = n_local
        else:
            n_local, n_global, rp, ci, va = gen(rank, world)
            source = "synthetic"
        bounds = [g * n_local for g in range(world + 1)]
        nnz_global = None
    else:
        n_global, grp, gci, gva, source = synth.load_or_make(args.workload)
        bounds = cdist.partition_rows_by_nnz(grp, world)
        rp, ci, va = cdist.slice_rows(grp, gci, gva, bounds[rank], bounds[rank + 1])
        n_local = bounds[rank + 1] - bounds[rank]
        nnz_global = int(gci.size)
    nnz_local = int(ci.size)
    x_host = np.arange(n_global, dtype=np.float64) * 0.25 / n_global        # test_spmv.cpp operand, scaled
    x_slice = x_host[bounds[rank]:bounds[rank + 1]].copy()

    # ---- exchange (N > 1): in-kernel halo / halo pull over shared slices, or RCCL all-gather ------------
    exchange, peer, p2p_error, halo_frac, native = "none", None, None, 0.0, None
    ci_dev, n_cols_dev = ci, n_global
    if use_dist:
        from cask_amd import p2p
        ci_ext, halo_cols, halo_owner, halo_index = p2p.plan_halo(ci, bounds, rank)
        halo_frac = cx.all_reduce_scalar(halo_cols.size / max(n_global, 1), dist.ReduceOp.MAX)
        want = os.environ.get("CASK_BENCH_EXCHANGE", "auto")
        if os.environ.get("CASK_BENCH_NO_P2P"):
            want = "all_gather"
        if want == "auto":
            want = "all_gather" if halo_frac > HALO_FRACTION_FOR_ALLGATHER else "p2p"
        exchange = "all_gather"
        if want == "p2p":
            def gather_objects(obj):
                out = [None] * world
                dist.all_gather_object(out, obj)
                return out

            try:
                peer = p2p.PeerExchange(bounds, rank, world, halo_owner, halo_index, dev, gather_objects)
                peer.x_local.copy_(torch.from_numpy(x_slice).to(dev))
                cx.host_barrier()                                   # every slice is in place before anyone pulls
                peer.pull()
                torch.cuda.synchronize()
                # self-check against a halo computed from the formula for x (no collective involved)
                wanted = torch.from_numpy(x_host[halo_cols]).to(dev)
                ok = bool(torch.equal(peer.x_ext[n_local:], wanted))
            except Exception as e:  # noqa: BLE001 - setup is collective: it fails on every rank or none
                ok, p2p_error = False, repr(e)
            ok = cx.all_reduce_scalar(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5
            if ok:
                exchange, ci_dev, n_cols_dev = "p2p", ci_ext, n_local + peer.n_halo
            else:
                if rank == 0:
                    print(f"[bench] peer-to-peer exchange unavailable ({p2p_error or 'halo mismatch'}); "
                          "using the RCCL all-gather", file=sys.stderr)
                if peer is not None:
                    peer.close()
                    peer = None
    gather, push, push_error = None, None, None
    n_cols_alg = n_cols_dev
    if exchange == "all_gather":
        # the gathered vector has a padded stride (rank g's slice at g*S): one collective per product whatever the
        # partition; the block's columns are remapped to it here, once
        gather = cdist.ShardedSpmv(bounds, rank, world, None, dev)
        ci_dev, n_cols_dev = gather.pad_columns(ci), gather.n_full
        # ... or no collective at all: slices pushed peer to peer (cask_hip_push_allgather), checked against the formula
        # for x on every rank first; any rank that cannot sends every rank back to RCCL
        if os.environ.get("CASK_BENCH_EXCHANGE", "auto") in ("auto", "push") and not os.environ.get("CASK_BENCH_NO_P2P"):
            from cask_amd import p2p

            def gather_objects(obj):
                out = [None] * world
                dist.all_gather_object(out, obj)
                return out
            t_sc = time.perf_counter()
            try:
                push = p2p.PushExchange(rank, world, gather.S, dev, gather_objects)
                # first contact: 50 exchanges, the operand changes every time, every entry of the gathered vector is
                # checked against its formula on every rank (cask_amd/selfcheck.py)
                pos = torch.arange(gather.n_full, device=dev)
                owner, off = pos // gather.S, pos % gather.S
                sizes_t = torch.tensor(gather.sizes, device=dev)
                starts_t = torch.tensor(gather.bounds[:-1], device=dev)
                valid = off < sizes_t[owner]
                idx_all = torch.where(valid, starts_t[owner] + off, torch.zeros_like(pos))
                idx_own = torch.arange(bounds[rank], bounds[rank + 1], device=dev)
                ok, push_error = selfcheck.check_push_allgather(torch, push, n_local, idx_own, idx_all, valid, n_global)
            except Exception as e:  # noqa: BLE001 - construction is collective: raised on every rank or on none
                ok, push_error = False, repr(e)
            ok, why = selfcheck.agree(ok, push_error, lambda v: cx.all_reduce_scalar(v, dist.ReduceOp.MIN), gather_objects)
            selfchecks["push_allgather"] = "ok" if ok else f"fell back: {why}"
            phase("exchange self-check (push all-gather)", t_sc)
            if ok:
                exchange = "push"
            else:
                push_error = why
                if rank == 0:
                    print(f"[bench] push all-gather unavailable ({push_error or 'gathered vector mismatch'}); using RCCL",
                          file=sys.stderr)
                if push is not None:
                    for ptr in push.peers.values():
                        p2p.close_peer(ptr)
                    push.peers = {}
                    cx.host_barrier()
                    push.close()
                    push = None
    gen_seconds = time.perf_counter() - t_gen
    phase(f"{args.workload}: generate + partition + exchange set-up", t_gen)
    # bytes the local kernel must move: x entries = the columns this block can reference
    alg_bytes = synth.algorithmic_bytes(n_local, n_cols_alg, nnz_local)
    matrix_bytes = 12 * nnz_local + 4 * (n_local + 1)
    copies = args.copies or max(2, -(-2 * INFINITY_CACHE_BYTES // max(matrix_bytes, 1)) + 1)

    forced = capi.make_params(variant=args.variant or 0, lanes_per_row=args.lanes, tile_width=args.tile,
                              items_per_thread=args.items, wg_size=args.wg,
                              index16=int(os.environ.get("CASK_BENCH_INDEX16", "0")))   # development A/B only
    mats = []
    torch.cuda.synchronize()
    t_up = time.perf_counter()
    rp_t = torch.from_numpy(rp).to(dev)
    ci_t0, va_t0 = torch.from_numpy(ci_dev).to(dev), torch.from_numpy(va).to(dev)
    torch.cuda.synchronize()
    upload_seconds = time.perf_counter() - t_up                  # one copy of the CSR arrays, host -> HBM
    t_plan = time.perf_counter()
    mats.append(capi.CsrMatrix.from_device(n_local, n_cols_dev, rp_t, ci_t0, va_t0, forced))
    torch.cuda.synchronize()
    plan_seconds = time.perf_counter() - t_plan                  # launch plan of one handle (merge-path cuts, x tiles, slots)
    for _ in range(copies - 1):
        ci_t, va_t = torch.from_numpy(ci_dev).to(dev), torch.from_numpy(va).to(dev)
        mats.append(capi.CsrMatrix.from_device(n_local, n_cols_dev, rp_t, ci_t, va_t, forced))
    x_local = None
    if exchange == "p2p":
        x_in = peer.x_ext                                        # [own slice | halo]
        # (collective below: every rank enters or none -- a rank whose block happens to reference no remote column too)
        if not os.environ.get("CASK_BENCH_NO_FUSED_HALO") and cx.all_reduce_scalar(peer.n_halo, dist.ReduceOp.MAX) > 0:
            # Fold the exchange into the product kernel: the workgroups at a seam read the halo from the
            # peers' slices themselves (cask_hip_csr_set_halo_sources), a step is ONE launch.  Checked
            # against the pull path first; any rank that disagrees sends every rank back to the pull.
            # First contact (cask_amd/selfcheck.py): 50 products whose operand changes every time.  The reference
            # products come from a private operand built from the formula, with the plain kernel, BEFORE the halo
            # sources are attached; then every rank rewrites its shared slice per exchange and the attached kernel
            # must reproduce them bit for bit while the local halo copy holds NaN.
            t_sc = time.perf_counter()
            idx_own = torch.arange(bounds[rank], bounds[rank + 1], device=dev)
            idx_halo = torch.from_numpy(halo_cols).to(dev)
            # a handle with halo sources runs the MERGE family only (set_halo_sources re-plans an AUTO handle that
            # resolved to another one): the reference products must come from that family too, or their sums are not
            # the attached kernel's bit for bit and a healthy exchange "fails" (cask_amd/dist.py does the same)
            saved_params = None
            if mats[0].params.as_dict()["variant"] != "merge":
                saved_params = [m.params for m in mats]
                for m in mats:
                    m.set_params(capi.make_params(variant="merge"))
            refs = []
            for e in range(selfcheck.N_EXCHANGES):
                xe = torch.cat([selfcheck.operand(e, idx_own, n_global), selfcheck.operand(e, idx_halo, n_global)])
                yr = torch.empty(n_local, dtype=torch.float64, device=dev)
                mats[0].spmv_device(xe, yr)
                refs.append(yr)
            torch.cuda.synchronize()
            try:
                for m in mats:
                    peer.attach(m)
                peer.x_ext[n_local:].fill_(float("nan"))         # the kernel must not read the local halo copy
                fused_ok, p2p_error = selfcheck.check_fused_halo(
                    torch, lambda yy: mats[0].spmv_device(x_in, yy), lambda e: refs[e], peer.x_local, idx_own,
                    cx.host_barrier, n_global)
            except Exception as e:  # noqa: BLE001
                fused_ok, p2p_error = False, repr(e)
            del refs
            fused_ok, why = selfcheck.agree(fused_ok, p2p_error, lambda v: cx.all_reduce_scalar(v, dist.ReduceOp.MIN), gather_objects)
            selfchecks["in_kernel_halo"] = "ok" if fused_ok else f"fell back: {why}"
            peer.x_local.copy_(torch.from_numpy(x_slice).to(dev))    # the benchmark's operand again
            cx.host_barrier()
            phase("exchange self-check (in-kernel halo)", t_sc)
            if fused_ok:
                exchange = "p2p_fused"
            else:
                if rank == 0:
                    print(f"[bench] in-kernel halo failed its first-contact check ({why}); pulling", file=sys.stderr)
                for m in mats:
                    m.set_halo_sources(n_local, None)
                if saved_params is not None:                     # the pull path has no family constraint: the caller's point again
                    for m, prm in zip(mats, saved_params):
                        m.set_params(prm)
                peer.pull()
    elif exchange == "push":
        # the operand of the benchmark does not change: it sits in place in BOTH gathered vectors (slot [rank]), so an
        # exchange only stores to the peers -- what a solver whose update kernel writes the slice there would see
        own = []
        for _ in range(2):
            slot = push.own_slot()
            slot.zero_()
            slot[:n_local].copy_(torch.from_numpy(x_slice).to(dev))
            own.append(slot)
            x_in = push.allgather(slot)
        x_local = own[0][:n_local]
    elif exchange == "all_gather":
        x_local = gather.x_slot[:n_local]                        # the slice lives where the collective reads it
        x_local.copy_(torch.from_numpy(x_slice).to(dev))
        x_in = gather.x_full
        native = gather.native_comm()                            # the engine's own RCCL communicator (nccl backend), else None
        gather.gather_x(x_local)
    else:
        x_in = torch.from_numpy(x_host).to(dev)
    y = torch.zeros(n_local, dtype=torch.float64, device=dev)

    phase(f"{args.workload}: upload + plans of {copies} rotating copies", t_up)
    # ---- measured DSE (cold: on the rotating copies), best point left active --------
    t_tune = time.perf_counter()
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
    phase(f"{args.workload}: measured DSE", t_tune)
    t_timed = time.perf_counter()

    def step(i):
        if exchange == "push":
            mats[i % copies].spmv_device(push.allgather(push.own_slot()), y)   # one exchange launch + the product
            return
        if exchange == "p2p":
            peer.pull()                                          # remote loads over xGMI, on the launch stream
        elif exchange == "all_gather":
            if native is not None:
                native.allgather(gather.x_slot, x_in)            # ONE ncclAllGather (padded stride) on this stream
            else:
                gather.gather_x(x_local)                         # torch.distributed all_gather_into_tensor, in place
        mats[i % copies].spmv_device(x_in, y)

    # ---- warm-up (eager) -------------------------------------------------------
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    launch_mode = args.launch if exchange not in ("all_gather", "push") else "eager"
    if exchange == "push" and args.launch in ("auto", "graph") and args.steps % 2 == 0:
        # the push exchange is plain kernel launches: capturable.  Its two gathered vectors alternate per exchange and
        # the product's operand pointer is frozen into the graph, so a replay must return to the parity it started
        # with: an even number of steps
        launch_mode = "graph"
    if launch_mode == "auto":
        # measured (one box): graph 8.65 / 8.27 us per step at 20 / 1000 steps, sequence 8.47 / 8.51: a graph start
        # costs ~9 us even queued behind another replay, which only a long region amortises
        launch_mode = "sequence" if args.steps < 200 and exchange in ("none", "p2p_fused") else "graph"
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
            # The GPU leaves its idle power state only under sustained load, and the capture above is a pause.  Round 4
            # (tools/drift_probe.py, profiles/r04_drift.txt): under back-to-back load the step time of this kernel is
            # constant to 0.1 % -- after the first ~150 ms, which run 1.5-2 % slower (the shader clock climbs for
            # ~500 ms); rounds 1-3 pre-rolled 40 ms.  Pre-roll --preroll-ms (300) of untimed replays so that the timed
            # windows run at the steady state a solver loop sees.
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            preroll = int(min(5000, max(2, args.preroll_ms / max(e0.elapsed_time(e1), 1e-3))))
        except Exception as e:  # pragma: no cover
            print(f"[bench] graph capture failed ({e!r}); timing eager launches", file=sys.stderr)
            graph, launch_mode = None, "eager"
        if use_dist and exchange == "push":
            # every rank must issue the same number of exchanges: agree on the pre-roll (it comes from a local clock)
            preroll = int(cx.all_reduce_scalar(preroll, dist.ReduceOp.MAX))
    if use_dist and launch_mode == "graph":
        # ranks must agree on the launch mode only for reporting; the timed region has no collective in p2p mode
        launch_mode = "graph" if cx.all_reduce_scalar(1.0 if graph is not None else 0.0, dist.ReduceOp.MIN) > 0.5 \
            else "eager"
        if launch_mode == "eager":
            graph = None

    # ---- timed region: windows of exactly K steps ---------------------------------
    sequence = launch_mode == "sequence" and exchange in ("none", "p2p_fused")     # a step is exactly one launch

    def run_steps():
        if graph is not None:
            graph.replay()
        elif sequence:
            capi.spmv_sequence_device(mats, x_in, y, args.steps)
        else:
            for i in range(args.steps):
                step(i)
    # windows: what the arguments ask for, bounded to ~4 s of timed work -- agreed over the ranks from one probe of K steps
    # (an exchange that is host-staged in a dry run, or slow on first contact with a fabric, must not eat the run)
    cx.host_barrier()
    t_probe = time.perf_counter()
    run_steps()
    cx.host_barrier()
    t_probe = time.perf_counter() - t_probe
    if use_dist:
        t_probe = cx.all_reduce_scalar(t_probe, dist.ReduceOp.MAX)
    n_win = max(3, min(n_windows(args), int(4.0 / max(t_probe, 1e-6))))
    if sequence:                                                # the same sustained load in front of the region
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run_steps()
        e1.record()
        torch.cuda.synchronize()
        preroll = int(min(20000, max(2, args.preroll_ms / max(e0.elapsed_time(e1), 1e-3))))
    elif graph is None:                                         # eager steps (collectives at N > 1): the same count on every
        preroll = int(min(500, max(1, args.preroll_ms * 1e-3 / max(t_probe, 1e-6))))   # rank -- t_probe is the ranks' maximum
    for _ in range(preroll):
        run_steps()
    clocks_before = gpu_clocks(dev.index or 0)                  # the pre-roll is still running: clocks under load
    native_windows = None
    window_form = {"form": None, "note": None}
    if sequence and not os.environ.get("CASK_BENCH_TORCH_EVENTS"):
        def native_windows(r):
            # stream launches + one event per window (the default); CASK_BENCH_WINDOW_FORM=graph: the K launches of every
            # window as kernel nodes of ONE graph with event-record nodes at the window boundaries (cask_hip_spmv_windows_device,
            # as_graph: native HIP -- torch refuses external events in a capture on ROCm).  Interleaved on one box the two
            # forms are level within the run-to-run noise (profiles/r04_launch_forms.txt).
            if os.environ.get("CASK_BENCH_WINDOW_FORM", "stream") == "graph":
                try:
                    u = capi.spmv_windows_device(mats, x_in, y, args.steps, r, as_graph=True)
                    window_form["form"] = "graph nodes + event-record nodes (one graph, launched twice: lead-in + timed)"
                    return u
                except capi.CaskHipError as e:
                    window_form["note"] = repr(e)
            window_form["form"] = "stream launches + events"
            return capi.spmv_windows_device(mats, x_in, y, args.steps, r, as_graph=False)
    tw = timed_windows(cx, run_steps, lead_in=graph.replay if graph is not None else (run_steps if sequence else None),
                       windows=n_win, native=native_windows)
    clocks_after = gpu_clocks(dev.index or 0)
    dev_ms, wall = tw["dev_ms"], tw["wall"] / tw["windows"]
    clock = (f"MEDIAN of {tw['windows']} back-to-back windows of K = {args.steps} steps, each between two HIP events on the "
             "launch stream (max over ranks of the per-rank median)") + (
        "; an untimed replay is queued in front of the first event so that the first window starts on a busy stream"
        if graph is not None else
        "; K launches + one timing event (no system-scope fence) per window from one C call, an untimed sequence of K "
        "launches queued in front of the first event" if sequence else "")
    phase(f"{args.workload}: warm-up, pre-roll and {tw['windows']} timed windows", t_timed)
    y_gpu = y.cpu().numpy()
    # post-run check: every rank's whole block against the CPU oracle (the global x is a formula, nothing to gather)
    import oracle
    bad, _ = oracle.mismatches(y_gpu, oracle.csr_spmv(rp, ci, va, x_host))
    if use_dist:
        nnz_total = cx.all_reduce_scalar(nnz_local, dist.ReduceOp.SUM)
        rows_wrong = int(cx.all_reduce_scalar(bad, dist.ReduceOp.SUM))
    else:
        nnz_total = float(nnz_local)
        rows_wrong = int(bad)

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

    # ---- the function the reference's clients call: Spmv::spmv(const Vector&) = cask_hip_spmv with HOST vectors (x up, one
    # launch, y down, per call; the matrix resident) -- never `value`, reported next to it (r6)
    host_entry = None
    if rank == 0 and world == 1:
        try:
            import ctypes
            hx, hy = np.ascontiguousarray(x_host[:mats[0].n_cols]), np.zeros(mats[0].n_rows)
            lib, px, py = capi.load(), hx.ctypes.data_as(ctypes.c_void_p), hy.ctypes.data_as(ctypes.c_void_p)
            host_entry = {}
            # (bench.py pins its own main thread for the MKL baseline -- OMP_PROC_BIND binds it when an OpenMP runtime starts;
            # a client's thread is not pinned, so the pin is lifted for this measurement and put back afterwards)
            pinned = os.sched_getaffinity(0)
            try:
                os.sched_setaffinity(0, range(os.cpu_count() or 1))
            except OSError:
                pass
            host_entry["caller_cpus"] = len(os.sched_getaffinity(0))
            for mode in ("pageable", "auto"):
                prev = capi.host_entry_mode(mode)
                for _ in range(10):
                    lib.cask_hip_spmv(mats[0]._h, px, py)
                best = float("inf")
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(40):
                        lib.cask_hip_spmv(mats[0]._h, px, py)
                    best = min(best, (time.perf_counter() - t0) / 40)
                capi.host_entry_mode(prev)
                host_entry[mode + "_usec"] = round(best * 1e6, 1)
            try:
                os.sched_setaffinity(0, pinned)
            except OSError:
                pass
            host_entry["vector_bytes_each_way"] = int(8 * mats[0].n_cols)
            host_entry["note"] = ("cask_hip_spmv per call, best of 5 loops of 40, x and y allocated once: `pageable` = hipMemcpyAsync "
                                  "from / to the caller's memory (ABI <= 6), `auto` = the default (64 KiB .. 4 MiB of vectors: threaded "
                                  "copies through pinned staging + a pull kernel + y written into pinned host memory); PCIe-inclusive, "
                                  "not `value`")
        except Exception as e:  # noqa: BLE001 - reporting only
            host_entry = {"error": repr(e)}

    rec = None
    if rank == 0:
        step_us = dev_ms * 1e3 / args.steps                # ONE clock: HIP events on the launch stream, max over ranks
        gflops = 2.0 * nnz_total / step_us * 1e-3
        achieved = alg_bytes / (step_us * 1e-6) / 1e9      # this rank's launch: its algorithmic bytes / the step time
        traffic, traffic_source = traffic_record(args.workload, design) if world == 1 else (None, None)
        like = {"cant": "cant-like", "cant3": "cant-like (3x3 node blocks, non-uniform band)",
                "webbase2": "webbase-1M-like (power-law in-degree: hub columns, site-block locality)"}.get(args.workload, args.workload + "-like")
        rec = {
            "metric": f"SpMV GFLOP/s (fp64 CSR, 2*nnz/t), SuiteSparse {like}", "value": round(gflops, 2),
            "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_us * 1e-3, 6), "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if source == "synthetic" else source,
            "host_wall_ms_per_step": round(wall * 1e3 / args.steps, 6), "clock": clock,
            **window_fields(tw, args.steps),
            "gpu_clocks_mhz": {"before_under_preroll": clocks_before, "during_windows": tw["clocks_during"],
                               "after": clocks_after},
            "config": {"workload": f"{like} CSR SpMV, {n_local} rows x {n_global} cols on rank 0, "
                                   f"{nnz_local} nnz on rank 0, x_i = 0.25 i / n",
                       "rows": n_global, "nnz": int(nnz_total),
                       "parallelism": f"row-blocks x{world}" + ("" if weak else " (nnz-balanced, one global matrix)"),
                       "exchange": {"none": "none",
                                    "p2p_fused": f"inside the product kernel: its seam workgroups load {peer.n_halo if peer else 0} "
                                                 "halo entries over xGMI from the neighbours' shared x slices "
                                                 "(one launch per step, no collective)",
                                    "p2p": f"per step: pull of {peer.n_halo if peer else 0} halo entries over xGMI from the "
                                           "neighbours' shared x slices (one kernel, no collective)",
                                    "push": "per step: every rank stores its x slice into every peer's gathered vector "
                                            "over xGMI + one flag per peer (cask_hip_push_allgather: one launch, no collective)",
                                    "all_gather": "per step: RCCL all_gather(x), padded stride: one collective" + (
                                        " issued by the engine (cask_hip_rccl_allgather)" if native is not None else "")}[exchange],
                       "halo_fraction_max": round(halo_frac, 4) if use_dist else None,
                       "rccl": cx.rccl,
                       "xgmi": xgmi_fields(exchange, world, step_us, stride=gather.S if gather is not None else 0,
                                           halo_by_owner=np.bincount(halo_owner, minlength=world).tolist() if peer is not None else None),
                       "exchange_selfcheck": selfchecks or None,
                       "rows_wrong_vs_oracle_all_ranks": rows_wrong,
                       "matrix_copies_rotated": copies, "launch": launch_mode, "window_form": window_form["form"],
                       "window_form_note": window_form["note"], "untimed_preroll_replays": preroll, "design_point": design,
                       "grid": info.grid, "lds_bytes": info.lds_bytes, "tune": tune_info,
                       "upload_seconds": round(upload_seconds, 4), "plan_seconds": round(plan_seconds, 4),
                       "generate_seconds": round(gen_seconds, 2)},
            "hbm_gbs_algorithmic": round(achieved * world, 1),
            "hbm_pct_of_peak": round(100.0 * achieved / HBM_PEAK_GBS, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "k_spmv_" + design["variant"], "algorithmic_bytes_per_launch": alg_bytes,
                         "launch_usec": round(step_us, 3)},
            "host_entry_usec": (host_entry or {}).get("auto_usec"), "host_entry": host_entry,
            "warm_cache": {"usec_graph": round(warm_graph, 3) if warm_graph else None,
                           "gflops_graph": round(2.0 * nnz_local / warm_graph * 1e-3, 2) if warm_graph else None,
                           "usec_eager_median": round(warm_med, 3), "usec_eager_min": round(warm_min, 3),
                           "note": "ONE copy of the matrix replayed (what a solver iteration sees; the Infinity Cache "
                                   "holds it, not an HBM figure): usec_graph is measured like `value` (graph of 200 "
                                   "launches), usec_eager_* are single launches between their own event pairs"},
        }
        if not args.no_cpu_baseline:
            if weak or world == 1:
                rec["cpu_baseline"] = cpu_baseline(rp, ci, va, x_host, y_gpu if world == 1 else None, args.cpu_seconds,
                                                   quick=getattr(args, "cpu_quick", False))
            else:
                rec["cpu_baseline"] = cpu_baseline(grp, gci, gva, x_host, None, args.cpu_seconds)
        else:
            rec["cpu_baseline"] = None
    if use_dist:
        cx.host_barrier()
        if push is not None:
            from cask_amd import p2p
            push.check()
            for ptr in push.peers.values():
                p2p.close_peer(ptr)
            push.peers = {}
            cx.host_barrier()                                       # owners free only after every peer has unmapped
            push.close()
        if peer is not None:
            from cask_amd import p2p
            for ptr in peer.peers.values():
                p2p.close_peer(ptr)
            peer.peers = {}
            cx.host_barrier()                                       # owners free only after every peer has unmapped
            peer.close()
    return rec


def run_solver(cx):
    """CG / BiCG passes (BASELINE configs[2], [4]); row-sharded for N > 1."""
    import torch
    import torch.distributed as dist
    from cask_amd import capi, synth
    from cask_amd import dist as cdist
    args, rank, world, dev, use_dist = cx.args, cx.rank, cx.world, cx.dev, cx.use_dist
    kind = args.solver
    name = args.workload
    t_all = time.perf_counter()
    n, rp, ci, va, source = synth.load_or_make(name)
    nnz = int(ci.size)
    x_true = np.random.default_rng(5).uniform(-1, 1, n)          # b = A x_true: the reference harness (test_utils.hpp:61-70)
    import scipy.sparse as sp
    b = np.asarray(sp.csr_matrix((va, ci, rp), shape=(n, n)) @ x_true)      # set-up, not the checker's job

    forced = capi.make_params(variant=args.variant or 0, tile_width=args.tile, items_per_thread=args.items, wg_size=args.wg)
    bounds = cdist.partition_rows_by_nnz(rp, world)
    halo_frac, exchange = 0.0, "none"
    selfchecks = {}
    sh = sht = None
    trp = tci = tva = None
    if kind == "bicg":
        trp, tci, tva = cdist.transpose_csr(n, n, rp, ci, va)
    if world > 1 or use_dist:                                       # (one forced rank: the sharded code path at world 1)
        from cask_amd import p2p
        lci = cdist.slice_rows(rp, ci, va, bounds[rank], bounds[rank + 1])[1]
        _, halo_cols, halo_owner_s, _ = p2p.plan_halo(lci, bounds, rank)
        halo_frac = cx.all_reduce_scalar(halo_cols.size / n, dist.ReduceOp.MAX)
        want = os.environ.get("CASK_BENCH_EXCHANGE", "auto")
        if want == "auto":
            want = "all_gather" if halo_frac > HALO_FRACTION_FOR_ALLGATHER else "p2p"
        exchange = "all_gather"
        if want == "p2p":
            fence = None
            if cx.backend != "nccl":                                 # host-staged control plane: fence on the host
                def fence():
                    torch.cuda.synchronize()
                    dist.barrier()
            try:
                from cask_amd import selfcheck
                t_sc = time.perf_counter()
                sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, params=forced, exchange="p2p", fence=fence,
                                                   fused_halo=True, solver_slots=6 if kind == "bicg" else 3, bounds=bounds,
                                                   selfcheck=selfcheck.N_EXCHANGES)
                selfchecks.update(getattr(sh, "selfcheck", None) or {})
                cx.phase("sharded operator + exchange self-check (in-kernel halo)", t_sc)
                if kind == "bicg":
                    sht = cdist.ShardedSpmv.from_global(trp, tci, tva, n, rank, world, params=forced, exchange="p2p",
                                                        fence=fence, fused_halo=True, share_with=sh)
                exchange = "p2p_fused"
            except Exception as e:  # noqa: BLE001 - collective: raised on every rank or none
                if rank == 0:
                    print(f"[bench] in-kernel halos unavailable ({e!r}); all-gathering the operands", file=sys.stderr)
                selfchecks.setdefault("in_kernel_halo", f"fell back: {e}")
                sh = sht = None
    if sh is None:
        sh = cdist.ShardedSpmv.from_global(rp, ci, va, n, rank, world, params=forced, bounds=bounds)
        if kind == "bicg":
            sht = cdist.ShardedSpmv.from_global(trp, tci, tva, n, rank, world, params=forced, bounds=bounds)
    b0, b1 = bounds[rank], bounds[rank + 1]
    bl = torch.from_numpy(b[b0:b1].copy()).to(dev)

    def solve(maxiters, tol):
        if kind == "bicg":
            return sh.bicg(sht, bl, maxiters=maxiters, tol=tol)
        return sh.cg(bl, maxiters=maxiters, tol=tol)

    cx.phase(f"{name} --solver {kind}: generate + sharded operators", t_all)
    # ---- the real solve, checked against the oracle ---------------------------------
    t_chk = time.perf_counter()
    xs, it, conv = solve(2000, 1e-5)
    torch.cuda.synchronize()
    x_all = xs.cpu().numpy()
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, x_all)
        x_all = np.concatenate(parts)
    check = None
    if rank == 0:
        import oracle                                               # post-run check only
        res = float(np.linalg.norm(b - oracle.csr_spmv(rp, ci, va, x_all)))
        want_x, want_it, want_conv = (oracle.cg_full if kind == "cg" else oracle.bicg)(rp, ci, va, b)
        check = {"iterations": it, "converged": conv, "oracle_iterations": want_it, "oracle_converged": want_conv,
                 "residual_2norm_by_oracle_product": res, "max_abs_diff_vs_oracle_solution": float(np.abs(x_all - want_x).max())}
    cx.phase(f"{name} --solver {kind}: the real solve + oracle check (rank 0)", t_chk)
    # ---- opt-in: the same solve over each route the dot products / operands can take (every rank reads the same variable)
    routes = None
    if os.environ.get("CASK_BENCH_COLLECTIVE_ROUTES") and (world > 1 or use_dist):
        t_rt = time.perf_counter()
        routes = {}
        for route in ("native", "torch", "peer"):
            sh.collectives_route = route
            xr, it_r, conv_r = solve(2000, 1e-5)
            torch.cuda.synchronize()
            routes[route] = {"collectives": sh.last_collectives, "iterations": it_r, "converged": conv_r,
                             "max_abs_diff_vs_first_solve": float((xr - xs).abs().max())}
        sh.collectives_route = None
        cx.phase(f"{name} --solver {kind}: the solve again over {len(routes)} collective routes", t_rt)
    # ---- warm-up + timed region: exactly K passes (tol = 0 never converges) -----------
    t_timed = time.perf_counter()
    if args.warmup:
        solve(args.warmup, 0.0)
    # windows: as many as the arguments ask for, but no more than fit ~4 s of solves -- agreed over the ranks from one
    # probe solve (a dry run whose all-reduces are host-staged takes milliseconds per pass; so might a first contact with
    # a slow fabric); never fewer than 3
    cx.host_barrier()
    t_probe = time.perf_counter()
    solve(args.steps, 0.0)
    cx.host_barrier()
    t_probe = cx.all_reduce_scalar(time.perf_counter() - t_probe, dist.ReduceOp.MAX) if use_dist else time.perf_counter() - t_probe
    windows = max(3, min(n_windows(args), int(4.0 / max(t_probe, 1e-6))))
    tw = timed_windows(cx, lambda: solve(args.steps, 0.0), windows=windows)
    dev_ms, wall = tw["dev_ms"], tw["wall"] / tw["windows"]
    cx.phase(f"{name} --solver {kind}: warm-up + {tw['windows']} timed solves of {args.steps} passes", t_timed)
    rec = None
    if rank == 0:
        step_us = dev_ms * 1e3 / args.steps
        b_spmv = synth.algorithmic_bytes(n, n, nnz)
        b_it = b_spmv + 96 * n if kind == "cg" else 2 * b_spmv + 152 * n      # SURVEY 8(d), unfused algorithmic bytes
        f_it = 2 * nnz + 12 * n if kind == "cg" else 4 * nnz + 20 * n
        achieved = b_it / world / (step_us * 1e-6) / 1e9                       # per GPU
        traffic, traffic_source = traffic_record(f"{name}_{kind}") if world == 1 else (None, None)
        # what a pass touches: the plan's streams (values + packed slots + row offsets) of A (and A^T) + the solver vectors
        ws = (1 if kind == "cg" else 2) * (9.5 * nnz + 4 * n) + (5 if kind == "cg" else 9) * 8 * n
        working_set_note = (f"a pass touches ~{ws / 2**20:.0f} MiB per GPU-set (plan streams + solver vectors) against the 256 MiB "
                            "Infinity Cache: passes replay the same data, so part of it is served by that cache, not HBM; "
                            "`frac` is on the UNFUSED algorithmic bytes of SURVEY 8(d) (fused kernels move fewer), `traffic` "
                            "is what the L2 asked the fabric for per pass (cache hits included)")
        rec = {
            "metric": f"{kind.upper()} pass GFLOP/s (fp64, {'2 nnz + 12 n' if kind == 'cg' else '4 nnz + 20 n'} flop per pass), "
                      f"SuiteSparse {name}-like",
            "value": round(f_it / step_us * 1e-3, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_us * 1e-3, 6), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if source == "synthetic" else source,
            "host_wall_ms_per_step": round(wall * 1e3 / args.steps, 6),
            "clock": f"MEDIAN of {tw['windows']} back-to-back solves of exactly K = {args.steps} passes, each between two HIP events",
            **window_fields(tw, args.steps),
            "config": {"workload": f"{kind} on the {name}-like system, {n} rows, {nnz} nnz, b = A x0, one step = one pass",
                       "rows": n, "nnz": nnz, "parallelism": f"row-blocks x{world} (nnz-balanced)",
                       "exchange": {"none": "none", "p2p_fused": "halos read inside the product kernels over xGMI; dot products: "
                                    "all-reduced device scalars (2 reductions per pass: config.collectives says how)",
                                    "all_gather": "per product: RCCL all_gather of the operand (padded stride); dot products all-reduced"}[exchange],
                       "halo_fraction_max": round(halo_frac, 4) if world > 1 else None,
                       "rccl": cx.rccl,
                       # per PRODUCT (a CG pass has one, a BiCG pass two) + 8-16 bytes per all-reduced dot
                       "xgmi": xgmi_fields(exchange, world, step_us, stride=getattr(sh, "S", 0),
                                           halo_by_owner=np.bincount(halo_owner_s, minlength=world).tolist()
                                           if (world > 1 or use_dist) and exchange != "all_gather" else None),
                       "pass_form": getattr(sh, "last_pass_form", None),
                       "exchange_selfcheck": {**selfchecks, **(getattr(sh, "selfcheck", None) or {})} or None,
                       "solve_check": check, "design_point": sh.matrix.params.as_dict(),
                       "engine_usec_per_pass_last_solve": round(sh.last_usec_per_iteration, 3),
                       "collectives": getattr(sh, "last_collectives", "none"),
                       "collective_routes": routes},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "working_set_note": working_set_note,
                         "kernel": f"{kind} pass (k_spmv_merge + update kernels)",
                         "algorithmic_bytes_per_launch": b_it // world, "launch_usec": round(step_us, 3)},
        }
        rec["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline_solver(kind, rp, ci, va, b, args.cpu_seconds,
                                                                                    quick=getattr(args, "cpu_quick", False))
    if world > 1 or use_dist:
        cx.host_barrier()
        if sht is not None and sht.exchange is sh.exchange:
            sht.exchange = None
        sh.close()
    return rec


if __name__ == "__main__":
    main()
