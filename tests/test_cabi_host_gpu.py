"""The C ABI from a plain C++ host (no Python / torch in the process): builds
tests/cabi_host/host.cpp against include/s4g_ops.h, libs4g_hip.so and the oracle library,
runs it on the GPU box and expects its self-check to pass."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_runs_the_c_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    lib_dir = os.path.join(ROOT, "s4g_release_amd")
    ora_dir = os.path.join(ROOT, "oracle")
    assert os.path.exists(os.path.join(lib_dir, "libs4g_hip.so")), "build the HIP library first"
    if not os.path.exists(os.path.join(ora_dir, "libs4g_oracle.so")):
        subprocess.run(["make", "-C", ora_dir], check=True, capture_output=True)
    exe = str(tmp_path / "cabi_host")
    cmd = [hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi_host", "host.cpp"), "-o", exe,
           "-L", lib_dir, "-ls4g_hip", "-L", ora_dir, "-ls4g_oracle",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + ora_dir]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    assert "cabi_host OK" in r.stdout
