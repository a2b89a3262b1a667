#!/usr/bin/env python3
"""Where a step of fps_pruned_kernel goes (debug build: make -C s4g_release_amd/csrc OBJDIR=... LIB=...
HIPFLAGS_EXTRA=-DS4G_FPS_STAMPS, loaded through S4G_HIP_LIB): per-wave s_memtime accumulators of the
three phases of an exchange -- [bound tests + touched-slot updates], [candidate search], [exchange incl.
the barrier wait] -- plus exchanges, picks and touched slots.  Usage: python tools/fps_stamps.py [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import _cabi, functions as F, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    pts = torch.from_numpy(synth.make_batch(list(range(B)), 25600)).to(dev)
    lib = _cabi.lib()
    lib.s4g_debug_fps_stamps.restype = ctypes.c_int
    lib.s4g_debug_fps_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    F.farthest_point_sample(pts, 5120)
    torch.cuda.synchronize()
    buf = np.zeros((64, 8, 8), dtype=np.uint64)
    lib.s4g_debug_fps_stamps(buf.ctypes.data, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    F.farthest_point_sample(pts, 5120)
    e1.record()
    torch.cuda.synchronize()
    lib.s4g_debug_fps_stamps(buf.ctypes.data, 1)
    a = buf[:B].astype(np.float64)
    ex, picks = a[:, 0, 3].mean(), a[:, 0, 4].mean()
    print("B=%d: %.3f ms per call; pruned phase: %.0f exchanges, %.0f picks (%.2f per exchange) per scene" % (
        B, e0.elapsed_time(e1), ex, picks, picks / ex))
    ph = a[:, :, :3] / a[:, :, 3:4]                      # cycles per exchange, per (scene, wave)
    tot = ph.sum(axis=2)
    print("cycles per exchange (mean over scenes and waves): update %.0f | candidate %.0f | exchange+wait %.0f | sum %.0f" % (
        ph[:, :, 0].mean(), ph[:, :, 1].mean(), ph[:, :, 2].mean(), tot.mean()))
    print("  slowest wave's update + candidate per exchange (what the barrier waits for): approx %.0f (mean over waves %.0f)" % (
        (ph[:, :, 0] + ph[:, :, 1]).max(axis=1).mean(), (ph[:, :, 0] + ph[:, :, 1]).mean()))
    print("touched slots per pick, per wave: %.2f (all waves: %.1f)" % ((a[:, :, 5] / a[:, :, 4]).mean(), (a[:, :, 5].sum(axis=1) / a[:, 0, 4]).mean()))
    print("per pick: %.0f cycles = %.3f us at 2.4 GHz" % (tot.mean() * ex / picks, tot.mean() * ex / picks / 2400))


if __name__ == "__main__":
    main()
