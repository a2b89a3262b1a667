"""CPU tests of the drop-in boundary: the C-ABI library loads (no GPU needed)
and exports every symbol include/s4g_ops.h declares; the Python operator API
keeps the reference's names; and the product path has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "s4g_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(s4g_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = _declared_symbols()
    for required in ("s4g_fps_f32", "s4g_ball_query_f32", "s4g_group_points_f32",
                     "s4g_group_points_backward_f32", "s4g_gather_points_f32", "s4g_three_nn_f32",
                     "s4g_three_interpolate_f32", "s4g_three_interpolate_backward_f32"):
        assert required in syms


def test_library_exports_every_declared_symbol():
    from s4g_release_amd import _cabi
    assert os.path.exists(_cabi.LIB_PATH), "build libs4g_hip.so first (__graft_entry__.build())"
    L = ctypes.CDLL(_cabi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(L, name), "libs4g_hip.so does not export %s" % name
    # and the ctypes table covers the header one to one
    assert sorted(_cabi.SIGNATURES) == _declared_symbols()
    assert _cabi.lib().s4g_abi_version() == _cabi.S4G_ABI_VERSION
    assert _cabi.lib().s4g_error_string(-1)


def test_operator_api_names_match_reference():
    from s4g_release_amd import functions as F, pn2_ext
    for name in ("gather_points", "farthest_point_sample", "ball_query", "group_points",
                 "search_nn_distance", "feature_interpolate", "gather_knn", "query_and_group",
                 "furthest_point_sample", "three_nn", "three_interpolate"):
        assert callable(getattr(F, name))
    for name in ("ball_query", "group_points_forward", "group_points_backward",
                 "farthest_point_sample", "point_search", "interpolate_forward",
                 "interpolate_backward"):
        assert callable(getattr(pn2_ext, name))


def test_no_cpu_fallback():
    from s4g_release_amd import functions as F
    pts = torch.zeros(1, 3, 8)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        F.farthest_point_sample(pts, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        F.ball_query(pts, pts, 0.1, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        F.search_nn_distance(pts, pts, 3)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "s4g_release_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_measurement_build_with_the_kernel_variants_compiles(tmp_path):
    """`-DS4G_VARIANTS` (csrc/variants/*.inc: the measured-slower kernels kept for A/B runs) must keep compiling
    against the shipped sources -- a change of a shared struct once broke it unnoticed -- and must say what it is."""
    import ctypes
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / "libs4g_hip_variants.so"
    out = subprocess.run(["make", "-C", os.path.join(root, "s4g_release_amd", "csrc"), "-j8", "OBJDIR=%s" % tmp_path,
                          "LIB=%s" % lib, "HIPFLAGS_EXTRA=-DS4G_VARIANTS"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    import torch  # noqa: F401  (its HIP runtime first, as _cabi.lib does)
    h = ctypes.CDLL(str(lib))
    from s4g_release_amd import _cabi
    assert h.s4g_build_variants() == 1 and h.s4g_abi_version() == _cabi.S4G_ABI_VERSION
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_ab_knobs_are_ignored_without_the_master_switch(monkeypatch):
    """include/s4g_ops.h: the A/B / test knobs are read only with S4G_TEST_KNOBS=1 (tests/conftest.py sets it).  Host side
    (`_cabi.knob`) and library side (`s4g_test_knobs_enabled`, csrc/s4g_common.h s4g::knob) agree; the production surface
    is the seven variables the header lists and nothing else in the package reads the environment."""
    import re
    from s4g_release_amd import _cabi
    monkeypatch.setenv("S4G_GEMM_FUSE2", "0")
    assert _cabi.knob("S4G_GEMM_FUSE2", "1") == "0"
    monkeypatch.delenv("S4G_TEST_KNOBS")
    assert _cabi.knob("S4G_GEMM_FUSE2", "1") == "1" and _cabi.knob("S4G_GEMM_FUSE2") is None
    monkeypatch.setenv("S4G_TEST_KNOBS", "yes")                    # only the literal "1" enables
    assert _cabi.knob("S4G_GEMM_FUSE2", "1") == "1"
    L = ctypes.CDLL(_cabi.LIB_PATH)                                 # no compute call: the symbol and its answer
    L.s4g_test_knobs_enabled.restype = ctypes.c_int
    variants = ctypes.CDLL(_cabi.LIB_PATH).s4g_build_variants()
    assert L.s4g_test_knobs_enabled() == (1 if variants else 0)
    monkeypatch.setenv("S4G_TEST_KNOBS", "1")
    assert L.s4g_test_knobs_enabled() == 1
    # every direct environment read in the package is one of the production variables (or torch.distributed's own)
    allowed = {"S4G_TEST_KNOBS", "S4G_HIP_LIB", "S4G_GEMM_MODE", "S4G_DIST_MODE", "S4G_BACKWARD", "S4G_GEO_STREAMS",
               "S4G_DENSE_STREAMS", "NCCL_MAX_NCHANNELS", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR"}
    pkg = os.path.join(ROOT, "s4g_release_amd")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            for name in re.findall(r'os\.environ(?:\.get|\.setdefault)?[\(\[]\s*"([A-Z0-9_]+)"', src):
                assert name in allowed, (f, name)
    csrc = os.path.join(pkg, "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(csrc, f)).read()
            direct = re.findall(r'(?<![:\w])getenv\("([A-Z0-9_]+)"\)', src)
            assert set(direct) <= {"S4G_TEST_KNOBS"}, (f, direct)
