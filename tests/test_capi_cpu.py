"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/cask_hip.h declares, validates arguments, and refuses to compute
without a GPU (no fallback)."""
import re
import subprocess

import numpy as np
import pytest

from cask_amd import capi
from conftest import REPO, have_gpu


def header_symbols():
    text = (REPO / "include" / "cask_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cask_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert capi.LIB_PATH.exists(), "run `make` / __graft_entry__.build() first"
    assert str(capi.LIB_PATH).startswith(str(REPO))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    out = subprocess.run(["nm", "-D", "--defined-only", str(capi.LIB_PATH)], check=True,
                         capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (cask_hip_[a-z0-9_]+)", out))
    assert set(header_symbols()) <= exported
    lib = capi.load()
    for name in header_symbols():
        assert hasattr(lib, name)
    assert lib.cask_hip_abi_version() == 7


def test_dfe_compat_triple_is_exported():
    """include/cask_hip_dfe.h: the reference's device function triple (GeneratedImplSupport.hpp:31-49)."""
    text = re.sub(r"/\*.*?\*/", "", (REPO / "include" / "cask_hip_dfe.h").read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(cask_hip_dfe_[a-z_]+)\s*\(", text)))
    assert declared == ["cask_hip_dfe_dram_read", "cask_hip_dfe_dram_write", "cask_hip_dfe_reset", "cask_hip_dfe_run"]
    out = subprocess.run(["nm", "-D", "--defined-only", str(capi.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", out), name


def test_native_rccl_entry_points_are_exported_without_a_link_dependency():
    """include/cask_hip_rccl.h: the collectives of the sharded solvers issued by the engine; RCCL itself is opened at
    run time, so the library must not name it as a dependency."""
    text = re.sub(r"/\*.*?\*/", "", (REPO / "include" / "cask_hip_rccl.h").read_text(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(cask_hip_rccl_[a-z_]+)\s*\(", text)))
    assert declared == sorted(capi.RCCL_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", str(capi.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", out), name
    needed = subprocess.run(["readelf", "-d", str(capi.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()


def test_native_rccl_entry_points_validate_their_arguments():
    L = capi.NativeComm._lib()
    assert L.cask_hip_rccl_allreduce(None, 1, None, None) == 1              # CASK_HIP_ERR_INVALID: no communicator
    assert L.cask_hip_rccl_allgather(None, None, None, None) == 1
    assert L.cask_hip_rccl_unique_id(None) == 1
    import ctypes
    h = ctypes.c_void_p()
    assert L.cask_hip_rccl_comm_create(None, 0, 1, None, ctypes.byref(h)) == 1
    assert L.cask_hip_rccl_comm_destroy(None) == 0


def test_code_object_is_gfx950_only():
    blob = capi.LIB_PATH.read_bytes()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_struct_layouts_match_header():
    # sizes implied by include/cask_hip.h (all int32/int64/double, natural alignment)
    import ctypes
    assert ctypes.sizeof(capi.Params) == 36
    assert ctypes.sizeof(capi.CsrInfo) == 64
    assert ctypes.sizeof(capi.DeviceProps) == 128
    assert ctypes.sizeof(capi.TunePoint) == 80


def test_argument_validation_happens_before_any_device_work():
    rp = np.array([0, 2, 1], dtype=np.int32)           # decreasing row_ptr
    with pytest.raises(ValueError, match="non-decreasing"):
        capi.CsrMatrix.from_host(2, 2, rp, [0, 1], [1.0, 2.0])
    with pytest.raises(ValueError, match="row_ptr\\[n_rows\\]"):
        capi.CsrMatrix.from_host(2, 2, [0, 1, 1], [0, 1], [1.0, 2.0])
    with pytest.raises(ValueError, match="column index"):
        capi.CsrMatrix.from_host(2, 2, [0, 1, 2], [0, 5], [1.0, 2.0])
    with pytest.raises(ValueError):
        capi.CsrMatrix.from_host(2, 2, [0, 1], [0], [1.0])          # row_ptr too short


@pytest.mark.skipif(have_gpu(), reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_gpu():
    assert capi.device_count() == 0
    with pytest.raises(capi.CaskHipError, match="no HIP device|no CPU fallback"):
        capi.CsrMatrix.from_host(2, 2, [0, 1, 2], [0, 1], [1.0, 2.0])


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cask_amd/ or include/ may reference it."""
    offenders = []
    for root in (REPO / "cask_amd", REPO / "include"):
        for f in root.rglob("*"):
            if f.is_file() and f.suffix in (".py", ".hip", ".hpp", ".h", ".cpp", ".c"):
                text = f.read_text(errors="replace")
                if re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M) or "cask_oracle" in text \
                        or "oracle/" in text:
                    offenders.append(str(f))
    assert not offenders, offenders
