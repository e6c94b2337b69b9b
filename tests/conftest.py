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
