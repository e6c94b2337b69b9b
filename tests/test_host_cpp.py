"""C++ host surface (include/cask/*.hpp + libCaskHip.so + generated libSpmv_hip.so).
CPU part: unit tests against the reference's gtest known answers, link seams.
GPU part: the integration client over every reference fixture and the DSE executable."""
import json
import re
import os
import subprocess

import pytest

from conftest import REPO, golden_matrix_files

LIBDIR = REPO / "cask_amd" / "lib"


def make(*targets):
    subprocess.run(["make", "-C", str(REPO), "-s", *targets], check=True, capture_output=True)


def test_host_unit_tests(plain_mtx_dir):
    make("build/test_host")
    out = subprocess.run([str(REPO / "build" / "test_host"), str(plain_mtx_dir)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(r"\d+ checks, 0 failures", out.stdout)


def test_host_code_under_sanitizers(plain_mtx_dir, tmp_path):
    """`make asan` (SURVEY section 5 "sanitizers"; the reference has coverage flags only,
    CMakeLists.txt:3): everything that is HOST code under AddressSanitizer + UBSan (-fno-sanitize-recover) on the CPU --
    the host surface's unit tests, the MatrixMarket ingest, the threaded staging copy of the host-vector entry (also under
    ThreadSanitizer), the oracle's C restatement (the whole of tests/test_oracle.py
    against the instrumented library) and the launch planners: every plan of plan_host.hpp / trsv_lanes_plan.hpp built
    for the 43 reference fixtures and the five small synthetic families, invariants checked (tests/cpp/test_planners.cpp).
    No GPU-side sanitizer is involved."""
    import sys
    import numpy as np
    from cask_amd import synth
    make("asan")
    asan = REPO / "build" / "asan"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(asan / "test_host"), str(plain_mtx_dir)], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and re.search(r"\d+ checks, 0 failures", out.stdout), out.stdout[-2000:] + out.stderr[-4000:]
    out = subprocess.run([str(asan / "ingest_time"), str(plain_mtx_dir / "matrices" / "OPF_3754.mtx")], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    dumps = tmp_path / "dumps"
    dumps.mkdir()
    for name in synth.GENERATORS:
        n, rp, ci, va = synth.small(name)
        with open(dumps / f"{name}.csr", "wb") as f:
            np.array([n, n, ci.size], dtype=np.int32).tofile(f)
            rp.astype(np.int32).tofile(f)
            ci.astype(np.int32).tofile(f)
            va.astype(np.float64).tofile(f)
    out = subprocess.run([str(asan / "test_planners"), str(plain_mtx_dir), str(dumps)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-4000:]
    m = re.search(r"(\d+) matrices, (\d+) lane-group runs \((\d+) chunks\), (\d+) checks, 0 failures", out.stdout)
    assert m and int(m.group(1)) >= 48 and int(m.group(2)) >= 10 and int(m.group(4)) > 10_000_000, out.stdout[-500:]
    for exe in ("test_host_copy", "test_host_copy_tsan"):      # the threaded staging copy (host_copy.hpp): ASan / UBSan, then TSan
        out = subprocess.run([str(asan / exe)], capture_output=True, text=True, env=env)
        assert out.returncode == 0 and re.search(r"\d+ checks, 0 failures", out.stdout), exe + out.stdout[-1000:] + out.stderr[-3000:]
    # the oracle's restatement: the CPU suite's oracle tests against the instrumented library (python itself is not
    # instrumented: the sanitizer runtimes are preloaded; CPython's own arenas make a leak check meaningless here)
    pre = ":".join(subprocess.run(["gcc", f"-print-file-name={lib}"], capture_output=True, text=True, check=True).stdout.strip()
                   for lib in ("libasan.so", "libubsan.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", str(REPO / "tests" / "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, cwd=str(REPO),
                         env=dict(os.environ, LD_PRELOAD=pre, ASAN_OPTIONS="detect_leaks=0", CASK_ORACLE_LIB=str(asan / "libcask_oracle.so")))
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_generated_library_exports_the_loader_constructor():
    """The seam the reference's clients link against: the loader ctor, both variants."""
    make("cask_amd/lib/lib-generated/libSpmv_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", str(LIBDIR / "lib-generated" / "libSpmv_hip.so")],
                         check=True, capture_output=True, text=True).stdout
    assert "_ZN4cask7runtime24SpmvImplementationLoaderC1Ev" in out
    assert "_ZN4cask7runtime24SpmvImplementationLoaderC2Ev" in out
    assert "cask_hip_generated_design_point" in out


def test_host_library_exports_the_spmv_methods():
    make("cask_amd/lib/libCaskHip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", str(LIBDIR / "libCaskHip.so")], check=True,
                         capture_output=True, text=True).stdout
    # the symbols a client compiled against the reference headers would reference
    assert "_ZN4cask4spmv4Spmv10preprocessERKNS_9CsrMatrixE" in out
    assert "_ZN4cask4spmv4Spmv4spmvERKNS_6VectorE" in out


def test_gen_impl_from_dse_out(tmp_path):
    dse = {"best_architectures": [
        {"architecture_params": {"variant": "merge", "lanes_per_row": 16, "tile_width": 1024, "wg_size": 512,
                                 "items_per_thread": 4, "xcd_remap": 1, "nontemporal": 1, "index16": 1},
         "matrices": ["cant.mtx"]},
        {"architecture_params": {"variant": 2, "lanes_per_row": 1, "tile_width": -1, "wg_size": 256,
                                 "items_per_thread": 8, "xcd_remap": 1, "nontemporal": 1, "index16": -1},
         "matrices": ["G3_circuit.mtx"]}]}
    (tmp_path / "dse_out.json").write_text(json.dumps(dse))
    subprocess.run(["python3", str(REPO / "tools" / "gen_impl.py"), "--dse", str(tmp_path / "dse_out.json"),
                    "--out-dir", str(tmp_path)], check=True, capture_output=True)
    src = (tmp_path / "GeneratedImplementations.cpp").read_text()
    assert src.count("new GeneratedSpmvImplementation(") == 2
    assert "{2, 16, 1024, 512, 4, 1, 1, 1}" in src and "{2, 1, -1, 256, 8, 1, 1, -1}" in src
    assert (tmp_path / "libSpmv_hip.so").exists()


def test_gen_impl_dfe_compat_binds_the_reference_triple(tmp_path):
    """--dfe-compat: the loader registers Spmv_<id>/_dramWrite/_dramRead wrappers (cask.py:259-283 shape)."""
    subprocess.run(["python3", str(REPO / "tools" / "gen_impl.py"), "--dfe-compat", "--out-dir", str(tmp_path)],
                   check=True, capture_output=True)
    src = (tmp_path / "GeneratedImplementations.cpp").read_text()
    assert "new GeneratedSpmvImplementation(0, Spmv_0, Spmv_0_dramWrite, Spmv_0_dramRead, 2147483647, 2, 1024, 16, false, 2)" in src
    out = subprocess.run(["nm", "-D", str(tmp_path / "libSpmv_hip_dfe.so")], check=True, capture_output=True, text=True).stdout
    assert "_ZN4cask7runtime24SpmvImplementationLoaderC1Ev" in out and "U cask_hip_dfe_run" in out


@pytest.mark.gpu
def test_integration_client_over_reference_fixtures(plain_mtx_dir):
    """ctest -R hw of the reference (CMakeLists.txt:135-139): test_spmv_<target> <matrix> for every fixture.  The
    reference starts one process per matrix; here the client takes the whole list in ONE process (--all: one HIP
    initialisation instead of 43 -- 16 s instead of 222 s on a slow box), and the one-matrix form
    the reference's ctest uses is run once."""
    make("clients")
    exe = REPO / "build" / "test_spmv_hip"
    paths = [str(plain_mtx_dir / (key + ".mtx")) for key, _ in golden_matrix_files()]
    out = subprocess.run([str(exe), "--all", *paths], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("Test passed!") == len(paths), (out.stdout[-1200:], out.stderr[-300:])
    assert f"{len(paths)} of {len(paths)} matrices passed" in out.stdout
    assert out.stdout.count("Result  Gflops (actual)=") == len(paths)
    for p in paths:
        assert f"Matrix {p} ok" in out.stdout, p
    one = subprocess.run([str(exe), paths[0]], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and "Test passed!" in one.stdout and "All tests passed!" in one.stdout, one.stdout[-600:]


@pytest.mark.gpu
def test_preconditioning_client_known_answers(plain_mtx_dir):
    """test/MklLayer.cpp + the ILU tests of test/LinearSolvers.cpp through the C++ surface, on the GPU."""
    make("clients")
    out = subprocess.run([str(REPO / "build" / "test_precond_hip"), str(plain_mtx_dir / "systems")],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "Test passed!" in out.stdout, (out.stdout[-600:], out.stderr[-600:])


@pytest.mark.gpu
def test_env_override_selects_the_variant(plain_mtx_dir):
    """CASK_HIP_VARIANT reaches the engine through the unchanged client (the reference picks designs by implId): every
    family, one process (--variants sets the variable before each run; `!fpga` must be rejected), and once the way a
    user would -- the variable set in the environment of the one-matrix form, a bad value a non-zero exit."""
    make("clients")
    exe = REPO / "build" / "test_spmv_hip"
    mtx = str(plain_mtx_dir / "matrices" / "test_cage6.mtx")
    out = subprocess.run([str(exe), "--variants", "vector,merge,merge_wave,scan,slice,auto,!fpga", mtx], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "7 of 7 variants behaved" in out.stdout, (out.stdout[-900:], out.stderr[-300:])
    assert out.stdout.count("Test passed!") == 6 and "Variant fpga rejected as it must be" in out.stdout
    bad = subprocess.run([str(exe), mtx], capture_output=True, text=True, env=dict(os.environ, CASK_HIP_VARIANT="fpga"), timeout=300)
    assert bad.returncode != 0


@pytest.mark.gpu
def test_cask_context_clients(plain_mtx_dir):
    """test/ClientTestSpmv.cpp and test/ClientTestCg.cpp: the CaskContext facade end to end."""
    make("clients")
    out = subprocess.run([str(REPO / "build" / "test_context_hip"), str(plain_mtx_dir / "systems")],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "Test passed!" in out.stdout, (out.stdout[-800:], out.stderr[-600:])


@pytest.mark.gpu
def test_bicg_client_reference_protocol(plain_mtx_dir):
    """test/test_bicg.cpp: identity, 2I (100, 10000) and bfwb62 through DfeBiCgSolver on the GPU."""
    make("clients")
    out = subprocess.run([str(REPO / "build" / "test_bicg_hip"), str(plain_mtx_dir / "matrices" / "bfwb62.mtx")],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "Test passed!" in out.stdout, (out.stdout[-800:], out.stderr[-600:])


@pytest.mark.gpu
def test_dse_executable_writes_dse_out(plain_mtx_dir, tmp_path):
    make("build/main")
    out = subprocess.run([str(REPO / "build" / "main"), str(plain_mtx_dir / "benchmark"),
                          str(REPO / "cask_amd" / "csrc" / "host" / "params.json")], cwd=tmp_path,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-500:]
    doc = json.loads((tmp_path / "dse_out.json").read_text())
    assert len(doc["best_architectures"]) == 3
    for arch in doc["best_architectures"]:
        assert arch["measured_gflops"] > 0 and arch["points_evaluated"] > 10
        assert arch["architecture_params"]["variant"] in (1, 2, 3, 4)         # any family may win on a small matrix
        assert arch["measured_usec"] > 0 and arch["measured_usec_warm"] > 0 and arch["matrix_copies_rotated"] >= 1
    # and the generator accepts what the DSE wrote
    subprocess.run(["python3", str(REPO / "tools" / "gen_impl.py"), "--dse", str(tmp_path / "dse_out.json"),
                    "--out-dir", str(tmp_path)], check=True, capture_output=True)


@pytest.mark.gpu
def test_sharded_solver_client_in_plain_c():
    """INTEGRATION.md section 6 as a program: C ABI only (cask_hip.h + cask_hip_rccl.h), RCCL communicator at world 1,
    cask_hip_solve_device with the native all-reduce / all-gather callbacks, CG and BiCG."""
    make("build/test_sharded_solver_hip")
    out = subprocess.run([str(REPO / "build" / "test_sharded_solver_hip")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Test passed!" in out.stdout, (out.stdout[-800:], out.stderr[-800:])
    assert "CG:" in out.stdout and "BiCG:" in out.stdout
