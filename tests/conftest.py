import gzip
import json
import os
import shutil
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_matrix_files(sets=("matrices", "benchmark", "systems", "failing")):
    """[(key, path)] for every coordinate-format fixture; key = '<set>/<name>'."""
    out = []
    for s in sets:
        for f in sorted((GOLDEN / s).glob("*.mtx*")):
            name = f.name[:-7] if f.name.endswith(".mtx.gz") else f.name[:-4]
            if name.endswith("_b") or name.endswith("_sol"):
                continue
            out.append((f"{s}/{name}", f))
    return out


@pytest.fixture(scope="session")
def expected_y():
    with np.load(GOLDEN / "spmv_expected.npz") as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def known_answers():
    return json.loads((GOLDEN / "known_answers.json").read_text())


@pytest.fixture(scope="session")
def plain_mtx_dir(tmp_path_factory):
    """Fixtures with every .mtx.gz unpacked, for C++ readers that take plain paths."""
    root = tmp_path_factory.mktemp("mtx")
    for s in ("matrices", "benchmark", "systems", "failing"):
        (root / s).mkdir()
        for f in (GOLDEN / s).glob("*.mtx*"):
            if f.name.endswith(".gz"):
                with gzip.open(f, "rb") as fi, open(root / s / f.name[:-3], "wb") as fo:
                    shutil.copyfileobj(fi, fo)
            else:
                shutil.copyfile(f, root / s / f.name)
    return root


def have_gpu():
    if os.environ.get("CASK_FORCE_NO_GPU"):
        return False
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def spawn_collect(fn, args, world):
    """Start `world` rank processes running ``fn(rank, *args, out)`` and return what each put on ``out`` as
    ``(rank, result)``, by rank.  The results are read BEFORE the join (a child blocks in ``put`` until its data is
    read: vectors are larger than a pipe's buffer), without a manager process (one process and ~0.4 s less per launch)."""
    import queue
    import torch.multiprocessing as mp
    out = mp.get_context("spawn").Queue()
    ctx = mp.spawn(fn, args=(*args, out), nprocs=world, join=False)
    by_rank = {}
    while len(by_rank) < world:
        try:
            rank, result = out.get(timeout=2.0)
            by_rank[rank] = result
        except queue.Empty:
            if ctx.join(timeout=0):            # (raises what a failed rank raised)
                raise RuntimeError(f"ranks exited without a result: got {sorted(by_rank)} of {world}")
    while not ctx.join():
        pass
    return [by_rank[r] for r in range(world)]
