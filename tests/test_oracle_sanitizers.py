"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C oracle asan` compiles `s4g_oracle.c` for both scalar types with `-fsanitize=address,undefined
-fno-sanitize-recover=undefined`; the oracle's own test module (`tests/test_oracle.py`: every operator, the literal
512-thread FPS emulation, the double build, edge cases) then runs in a CHILD process that loads that library
(S4G_ORACLE_LIB) with libasan preloaded -- a heap overflow, use-after-free or UB (signed overflow, bad shift,
misaligned access, out-of-range float -> int) aborts the child.  Nothing here touches a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san_lib(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_suite_is_clean_under_asan_and_ubsan():
    asan = _san_lib("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "libs4g_oracle_asan.so")
    env = dict(os.environ, S4G_ORACLE_LIB=lib, LD_PRELOAD=asan, OMP_NUM_THREADS="4",
               # CPython itself "leaks" at exit and intercepts signals; everything else stays fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:handle_segv=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import oracle as O\n"
            "assert O._LIB_PATH.endswith('libs4g_oracle_asan.so'), O._LIB_PATH\n"
            "import ctypes; O.lib()\n"
            "import pytest\n"
            "sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', %r]))\n"
            % (ROOT, os.path.join(ROOT, "tests", "test_oracle.py")))
    proc = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=1500)
    tail = (proc.stdout[-3000:] + "\n" + proc.stderr[-3000:])
    assert proc.returncode == 0, tail
    assert "passed" in proc.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_the_sanitizer_build_really_catches_an_overflow():
    """The harness is live: an output buffer half the size the call writes, handed to the ASan build's gather,
    aborts the child with a heap-buffer-overflow report."""
    asan = _san_lib("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "libs4g_oracle_asan.so")
    env = dict(os.environ, S4G_ORACLE_LIB=lib, LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77:handle_segv=0")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, ctypes\n"
            "from oracle import oracle as O\n"
            "L = O.lib()\n"
            "pts = np.zeros((1, 3, 4096), np.float32); idx = np.zeros((1, 2048), np.int64)\n"
            "out = np.zeros((1, 3, 1024), np.float32)         # the call writes (1, 3, 2048)\n"
            "L.s4g_oracle_gather_points(O._fp(pts), O._ip(idx), 1, 3, 4096, 2048, O._fp(out))\n"
            "print('survived')\n" % ROOT)
    proc = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "survived" not in proc.stdout
    assert "AddressSanitizer" in proc.stderr and "heap-buffer-overflow" in proc.stderr, proc.stderr[-2000:]
