"""bench.py end to end on the GPU box: the one-rank line the driver records, and a two-rank dry run of
the N > 1 path (both ranks share the box's one GPU; control plane gloo) -- the exchange mode, the
per-rank oracle check and the JSON contract are exactly what an 8-GPU launch goes through."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO, have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs a GPU")]

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def run_bench(extra, env=None, world=1, timeout=900, launcher=True):
    cmd = [sys.executable]
    if "--solver" not in extra and "--workload" not in extra and "--with-others" not in extra:
        extra = extra + ["--no-others"]                     # the appended workloads have their own test below
    extra = [e for e in extra if e != "--with-others"]
    if "--preroll-ms" not in extra:
        extra = extra + ["--preroll-ms", "40"]              # (the 300 ms default is for measurements, not for these checks)
    if world > 1 and launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                "--master-port", str(free_port())]
    cmd += [str(REPO / "bench.py"), "--gpus", str(world)] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-3000:]
    return last_json_line(out.stdout)


SHARE = {"CASK_BENCH_SHARE_DEVICE": "1", "CASK_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"}


def test_ranks_that_share_a_device_are_refused_unless_declared():
    """N ranks on fewer than N devices is not a multi-GPU measurement: without
    CASK_BENCH_SHARE_DEVICE the run ends with exit status 4 before anything is timed (two ranks started by hand, both
    on device 0, control plane gloo)."""
    port = str(free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   CASK_BENCH_BACKEND="gloo")
        env.pop("CASK_BENCH_SHARE_DEVICE", None)
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                                       "--no-cpu-baseline", "--no-tune", "--no-others"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [4, 4], [(p.returncode, o[1][-500:]) for p, o in zip(procs, outs)]
    assert "refusing to time" in outs[0][1] and not any(line.startswith("{") for line in outs[0][0].splitlines())


def test_stale_halo_fault_sends_every_rank_to_the_fallback():
    """CASK_FAULT_STALE_HALO makes a rank serve the PREVIOUS operand on odd exchanges -- what a
    missed fence or a stale line looks like to its peers.  The self-check must see it on every path, every rank must
    take the fallback together, and the run must still be right."""
    # (a) in-kernel halo -> the halo pull        (--windows 3: these runs test the PROTOCOL; over the gloo control plane
    # of a shared-device dry run a collective fallback costs ~50 ms a step, and 49 windows of it took 28 s)
    rec = run_bench(["--steps", "10", "--warmup", "2", "--no-tune", "--copies", "2", "--no-cpu-baseline", "--windows", "3"],
                    dict(SHARE, CASK_FAULT_STALE_HALO="halo", CASK_SELFCHECK_EXCHANGES="12"), world=2)
    assert rec["config"]["exchange_selfcheck"]["in_kernel_halo"].startswith("fell back: rank ") and \
        "in-kernel halo:" in rec["config"]["exchange_selfcheck"]["in_kernel_halo"]          # every failing rank's reason
    assert rec["config"]["exchange"].startswith("per step: pull of"), rec["config"]["exchange"]
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0
    # (b) push all-gather -> the collective   ((b) and (c) test the PROTOCOL: an eighth of the rows; the full-size dry runs
    # of the same two workloads are test_config4_... / test_config5_... below)
    args = ["--steps", "10", "--warmup", "2", "--workload", "webbase-1M", "--no-tune", "--copies", "2", "--no-cpu-baseline",
            "--windows", "3"]
    rec = run_bench(args, dict(SHARE, CASK_FAULT_STALE_HALO="push", CASK_BENCH_SHRINK="8", CASK_SELFCHECK_EXCHANGES="12"), world=3)
    assert rec["config"]["exchange_selfcheck"]["push_allgather"].startswith("fell back: rank ") and \
        "push all-gather:" in rec["config"]["exchange_selfcheck"]["push_allgather"]
    assert rec["config"]["exchange"].startswith("per step: RCCL all_gather(x)")
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0
    # (c) sharded solver: in-kernel halos -> all-gathered operands, peer-store all-reduce -> the collective
    rec = run_bench(["--steps", "10", "--warmup", "2", "--workload", "atmosmodd", "--solver", "bicg", "--no-cpu-baseline", "--windows", "3"],
                    dict(SHARE, CASK_FAULT_STALE_HALO="1", CASK_PEER_ALLREDUCE="1", CASK_BENCH_SHRINK="8", CASK_SELFCHECK_EXCHANGES="12"),
                    world=2)
    sc = rec["config"]["exchange_selfcheck"]
    assert sc["in_kernel_halo"].startswith("fell back") and sc["peer_store_allreduce"].startswith("fell back"), sc
    assert rec["config"]["exchange"].startswith("per product: RCCL all_gather")
    assert not rec["config"]["collectives"].startswith("peer-store")
    chk = rec["config"]["solve_check"]
    assert chk["converged"] and abs(chk["iterations"] - chk["oracle_iterations"]) <= 2


def test_rccl_collectives_run_at_world_one():
    """the nccl (= RCCL) backend itself -- init_process_group("nccl"), all_gather_into_tensor of x
    and all_reduce of scalars on the device -- with the one rank a 1-GPU box allows: the config-4 step (all-gather +
    product) and the config-5 solver pass (operand all-gather + all-reduced dots through the engine's callbacks)."""
    env = {"CASK_BENCH_FORCE_DIST": "1", "CASK_BENCH_EXCHANGE": "all_gather", "CASK_FORCE_COLLECTIVES": "1",
           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
           "CASK_BENCH_SHRINK": "8"}        # the RCCL calls are the point, not the size (full size: the dry runs below)
    rec = run_bench(["--steps", "10", "--warmup", "2", "--workload", "webbase-1M", "--no-tune", "--copies", "2",
                     "--no-cpu-baseline"], env)
    assert rec["config"]["exchange"].startswith("per step: RCCL all_gather(x)") and "issued by the engine" in rec["config"]["exchange"]
    rccl = rec["config"]["rccl"]                                  # what RCCL itself reports for the engine's communicator
    assert rccl["backend"].startswith("nccl") and rccl["comm_nranks"] == 1 and rccl["distinct_devices"] == 1 and rccl["errors"] is None
    assert rec["config"]["launch"] == "eager"
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0
    # the config-5 pass: one process, one RCCL start-up, the solve over each route its collectives can take
    env.update(MASTER_PORT=str(free_port()), CASK_BENCH_COLLECTIVE_ROUTES="1")
    rec = run_bench(["--steps", "10", "--warmup", "2", "--workload", "atmosmodd", "--solver", "bicg", "--no-cpu-baseline"], env)
    chk = rec["config"]["solve_check"]
    assert chk["converged"] and abs(chk["iterations"] - chk["oracle_iterations"]) <= 2
    assert chk["residual_2norm_by_oracle_product"] <= 2e-5
    assert rec["config"]["collectives"].startswith("native RCCL all-reduce")      # ncclAllReduce / ncclAllGather from the engine
    assert rec["config"]["collectives"].endswith("operand: native RCCL all-gather")
    routes = rec["config"]["collective_routes"]
    assert routes["native"]["collectives"] == rec["config"]["collectives"]
    # ... the same through the torch.distributed callbacks
    assert routes["torch"]["collectives"].startswith("torch.distributed all-reduce")
    assert routes["torch"]["collectives"].endswith("operand: torch.distributed all-gather")
    # ... and with the dot products reduced by peer stores (opt-in): no all-reduce call in a pass
    assert routes["peer"]["collectives"].startswith("peer-store all-reduce")
    assert rec["config"]["exchange_selfcheck"]["peer_store_allreduce"] == "ok"
    for r in routes.values():
        assert r["converged"] and r["iterations"] == chk["iterations"] and r["max_abs_diff_vs_first_solve"] <= 1e-9, routes


def test_config4_webbase_row_partitioned_dry_run():
    """BASELINE configs[3] as bench.py runs it on N GPUs, here 4 ranks sharing the GPU (gloo): one global matrix,
    nnz-balanced row blocks, the gathered x its halo fraction selects -- pushed peer to peer (the default) and through
    the collective (forced) -- every rank's rows against the oracle."""
    args = ["--steps", "10", "--warmup", "2", "--workload", "webbase-1M", "--no-tune", "--copies", "2", "--no-cpu-baseline"]
    rec = run_bench(args, SHARE, world=4)
    assert rec["n_gpus"] == 4 and rec["scaling"] == "strong" and rec["config"]["rows"] == 1_000_005
    assert rec["config"]["exchange"].startswith("per step: every rank stores its x slice"), rec["config"]["exchange"]
    assert rec["config"]["halo_fraction_max"] > 0.10
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0
    assert rec["config"]["exchange_selfcheck"] == {"push_allgather": "ok"}
    rec = run_bench(args, dict(SHARE, CASK_BENCH_EXCHANGE="all_gather"), world=4)
    assert rec["config"]["exchange"] == "per step: RCCL all_gather(x), padded stride: one collective"   # gloo dry run
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0


def test_config5_atmosmodd_bicg_sharded_dry_run():
    """BASELINE configs[4]: BiCG on the full atmosmodd-like system, A and A^T row-sharded over 4 ranks sharing the GPU,
    halos read in-kernel, dots all-reduced; iteration count and true residual against the oracle."""
    rec = run_bench(["--steps", "10", "--warmup", "2", "--workload", "atmosmodd", "--solver", "bicg", "--no-cpu-baseline"],
                    SHARE, world=4)
    assert rec["n_gpus"] == 4 and rec["config"]["exchange"].startswith("halos read inside the product kernels"), \
        (rec["config"]["exchange"], rec["config"].get("exchange_selfcheck"))
    assert rec["config"]["exchange_selfcheck"]["in_kernel_halo"] == "ok"
    chk = rec["config"]["solve_check"]
    assert chk["converged"] and chk["oracle_converged"] and abs(chk["iterations"] - chk["oracle_iterations"]) <= 2
    assert chk["residual_2norm_by_oracle_product"] <= 5e-5 and chk["max_abs_diff_vs_oracle_solution"] <= 1e-5


def test_default_line_carries_the_other_baseline_configs():
    """the N = 1 default run times cant3, webbase-1M, CG on G3_circuit and BiCG on atmosmodd after
    the headline and reports them under config.other_workloads with their own checks."""
    rec = run_bench(["--steps", "20", "--warmup", "4", "--copies", "3", "--cpu-seconds", "0.5", "--other-steps", "40",
                     "--with-others"])
    others = rec["config"]["other_workloads"]
    assert [o.get("error") for o in others] == [None] * 5, others
    assert "appended_workloads_failed" not in rec
    names = [o["workload"] for o in others]
    assert "cant" in names[0] and "webbase-1M" in names[1] and "hub columns" in names[2] and "G3_circuit" in names[3] \
        and "atmosmodd" in names[4]
    for o in others[:3]:
        assert o["rows_wrong"] == 0 and 0 < o["frac"] < 1 and o["usec"] > 0
        assert o["windows"] >= 31 and o["usec_p10"] <= o["usec"] <= o["usec_p90"]
    for o in others[3:]:
        chk = o["solve_check"]
        assert chk["converged"] and abs(chk["iterations"] - chk["oracle_iterations"]) <= 2
        assert chk["residual_2norm_by_oracle_product"] <= 5e-5 and 0 < o["frac"] < 1
        assert o["traffic"] and "profiles/traffic_" in o["traffic_source"]          # no "traffic": null
    # every appended workload carries its MKL column (a bounded sample: 16 pinned threads), and
    # BASELINE configs[2] at full size -- the CG line on G3_circuit -- is this run's fourth entry (r4 timed it in a run of its own)
    for o in others:
        cb = o["cpu_baseline"]
        assert cb["kind"] in ("mkl", "port") and cb["value"] > 0 and cb["unit"] == "GFLOP/s", o
    assert others[3]["value"] > 100 and others[3]["solve_check"]["residual_2norm_by_oracle_product"] <= 2e-5
    assert rec["cpu_baseline"]["kind"] in ("mkl", "port")
    # the headline is the cant line, on its own clock, unchanged by what follows
    assert rec["config"]["workload"].startswith("cant-like") and rec["steps"] == 20
    # ... and it is the one-rank line the driver records: the JSON contract, ONE clock, the CPU column, the windows
    assert CONTRACT_KEYS <= set(rec)
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["dtype"] == "f64" and rec["value"] > 100
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # ONE clock: value, ms_per_step and the roofline all derive from the same device-event time
    assert abs(rec["ms_per_step"] * 1e3 - roof["launch_usec"]) < 1e-2
    assert abs(rec["value"] - 2.0 * rec["config"]["nnz"] / roof["launch_usec"] * 1e-3) <= 0.01 * rec["value"]
    assert roof["traffic"] is None or "profiles/" in roof["traffic_source"]
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] in ("mkl", "port") and cpu["parity_gpu_vs_cpu_mismatches"] == 0 and cpu["cores"] >= 1
    if cpu["kind"] == "mkl":
        assert cpu["mismatches_vs_oracle"] == 0 and cpu["port"]["kind"] == "port" and cpu["port"]["cores"] == 1
    # ms_per_step is the MEDIAN of >= 31 back-to-back K-step windows; spread and clocks in the line
    assert rec["windows"] >= 31 and rec["ms_per_step_min"] <= rec["ms_per_step_p10"] <= rec["ms_per_step"] <= rec["ms_per_step_p90"] <= rec["ms_per_step_max"]
    assert "MEDIAN" in rec["clock"] and "gpu_clocks_mhz" in rec
    assert rec["config"]["plan_seconds"] > 0 and rec["config"]["upload_seconds"] > 0
    assert "workload" in rec["config"] and "model" not in rec["config"]


def test_two_rank_line_appends_the_strong_scaling_configs():
    """at N > 1 configs[3] (webbase-1M, all-gather) and configs[4] (BiCG on atmosmodd) follow the weak
    cant headline; two ranks sharing the GPU, started by bench.py itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SHARE)
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4", "--copies", "2",
           "--no-cpu-baseline", "--other-steps", "20", "--preroll-ms", "40"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = last_json_line(out.stdout)
    # `python bench.py --gpus 2` with WORLD_SIZE unset -- the form of the driver's recorded command --
    # starts its two ranks itself (child processes), relays rank 0's line (ONE line) and exits 0
    assert len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1
    # the weak-scaling headline at two ranks, WITH the DSE as the driver launches it: the blocks are tuned with their halo
    # sources attached, the halos are read inside the product kernel, every row of both ranks is right
    assert rec["config"]["tune"]["points"] >= 10            # blocks with halo sources: the MERGE family only
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["exchange"].startswith("inside the product kernel"), rec["config"]["exchange"]
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0
    assert rec["config"]["rows"] == 2 * 62451
    # the first-contact self-check ran (50 exchanges, the operand changes every time) and passed
    assert rec["config"]["exchange_selfcheck"] == {"in_kernel_halo": "ok"}
    # the line says what the collectives layer saw and what a step moves over xGMI
    rccl = rec["config"]["rccl"]
    assert rccl["backend"] == "gloo" and rccl["world_size_seen"] == 2 and len(rccl["devices"]) == 2
    assert rccl["distinct_devices"] == 1 and rccl["shared_device_dry_run"] is True       # both ranks on the box's one GPU, declared
    assert all(":" in d for d in rccl["devices"])                                          # host:PCI bus id per rank
    xg = rec["config"]["xgmi"]
    assert xg["bytes_received_per_step_per_gpu"] > 0 and xg["bytes_per_link_max"] > 0 and xg["gbs_per_link_at_step_time"] > 0
    others = rec["config"]["other_workloads"]
    assert [o.get("error") for o in others] == [None] * 3, others
    assert others[0]["rows_wrong"] == 0 and others[0]["scaling"] == "strong"
    for o in others[1:]:
        chk = o["solve_check"]
        assert chk["converged"] and abs(chk["iterations"] - chk["oracle_iterations"]) <= 2
