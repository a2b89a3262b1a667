#!/usr/bin/env python3
"""Phase timeline of mlp_chain_kernel from s_memtime stamps (debug build:
`make -C s4g_release_amd/csrc clean all HIPFLAGS_EXTRA=-DS4G_CHAIN_STAMPS`).  Runs one fused forward and,
after each chain launch of interest, reads the stamps of wave 0 of the first 512 workgroups.
Usage: python tools/chain_stamps.py [precision]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import _cabi, fused, synth  # noqa: E402
from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_  # noqa: E402

NAMES = ["start", "init issued", "panel in LDS", "barrier", "P0 strip", "P0 epi1", "P0 bar", "P0 epi2+w",
         "P1 strip", "P1 epi1", "P1 bar", "end", "S0 begin", "S0 strip", "S1 begin", "S1 strip"]


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
    dev = torch.device("cuda:0")
    torch.manual_seed(20260101)
    net = build_pointnet2_cls(S4GConfig())
    randomize_bn_(net, 20260102)
    net = net.to(dev).eval()
    run = fused.FusedPointNet2(net, precision=prec)
    pts = torch.from_numpy(synth.make_batch(list(range(16)), 25600)).to(dev)
    lib = _cabi.lib()
    lib.s4g_debug_chain_stamps.restype = ctypes.c_int
    lib.s4g_debug_chain_stamps.argtypes = [ctypes.c_void_p]
    orig = run._gemm

    def wrapped(name, *a, **kw):
        orig(name, *a, **kw)
        if kw.get("layer2") is not None:
            torch.cuda.synchronize()
            buf = np.zeros((512, 16), dtype=np.uint64)
            rc = lib.s4g_debug_chain_stamps(buf.ctypes.data)
            assert rc == 0
            st = buf.astype(np.int64)
            base = st[:, 0:1]
            rel = st - base
            order = [0, 1, 2, 3, 4, 5, 6, 7] + ([8, 9, 10] if st[:, 8].max() > 0 and (st[:, 8] > st[:, 7]).mean() > 0.5 else []) + [12, 13, 14, 15, 11]
            print("== %s (%s): median ticks since workgroup start, 512 workgroups from the middle of the grid" % (name, prec))
            prev = 0
            for i in order:
                ok = rel[:, i] > 0 if i else np.ones(512, bool)
                if not ok.any():
                    continue
                med = float(np.median(rel[ok, i]))
                print("   %-14s %8.0f   (+%6.0f)" % (NAMES[i], med, med - prev))
                prev = med
    run._gemm = wrapped
    with torch.no_grad():
        run({"scene_points": pts})
        run({"scene_points": pts})
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
