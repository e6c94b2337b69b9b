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


def test_single_rank_line():
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "20", "--warmup", "4", "--no-tune",
                          "--copies", "3", "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = last_json_line(out.stdout)
    assert CONTRACT_KEYS <= set(rec)
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["dtype"] == "f64" and rec["value"] > 100
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["parity_gpu_vs_cpu_mismatches"] == 0
    assert "workload" in rec["config"] and "model" not in rec["config"]


def test_two_rank_dry_run_reads_halos_in_kernel():
    env = dict(os.environ, CASK_BENCH_SHARE_DEVICE="1", CASK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), str(REPO / "bench.py"), "--gpus", "2", "--steps", "20",
           "--warmup", "4", "--no-tune", "--copies", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = last_json_line(out.stdout)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["exchange"].startswith("inside the product kernel"), rec["config"]["exchange"]
    assert rec["config"]["rows_wrong_vs_oracle_all_ranks"] == 0
    assert rec["config"]["rows"] == 2 * 62451
