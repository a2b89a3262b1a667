#!/usr/bin/env python3
"""Time-boxed randomized sweep of the deterministic backward scatters (csrc/scatter.hip) against the CPU oracle: BIT-EXACT,
random shapes around the kernels' block sizes, hot targets, out-of-range-free indices, extreme gradient magnitudes.
Usage on a GPU box:  python tools/fuzz_scatter.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                      # noqa: E402  (checker only)
from s4g_release_amd import functions as F          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    F.set_backward_mode("deterministic")
    t0, case = time.time(), 0
    while time.time() - t0 < a.seconds:
        case += 1
        B = int(rng.integers(1, 4))
        C = int(rng.choice([1, 3, 4, 7, 32, 65]))
        N = int(rng.choice([1, 2, 5, 255, 256, 257, 1000, 5120]))
        hot = rng.random() < 0.5
        if rng.random() < 0.5:
            M, K = int(rng.integers(1, 300)), int(rng.choice([1, 3, 16, 64]))
            idx = rng.integers(0, N, size=(B, M, K))
            if hot:
                idx[:, :, K // 2:] = idx[:, :, :1]
            g = (rng.standard_normal((B, C, M, K)) * np.exp(rng.uniform(-20, 20, size=(B, C, M, K)))).astype(np.float32)
            got = F._group_points_backward(torch.from_numpy(g).to(dev), torch.from_numpy(idx).to(dev), N).cpu().numpy()
            ref = O.group_points_backward(g, idx, N)
            tag = "group B=%d C=%d N=%d M=%d K=%d hot=%d" % (B, C, N, M, K, hot)
        else:
            N1 = int(rng.integers(1, 3000))
            N2 = max(N, 3)
            idx = rng.integers(0, N2, size=(B, N1, 3))
            if hot:
                idx[:, : N1 // 2] = idx[:, :1]
            w = rng.random((B, N1, 3), dtype=np.float32)
            g = (rng.standard_normal((B, C, N1)) * np.exp(rng.uniform(-20, 20, size=(B, C, N1)))).astype(np.float32)
            got = F._interpolate_backward(torch.from_numpy(g).to(dev), torch.from_numpy(idx).to(dev),
                                          torch.from_numpy(w).to(dev), N2).cpu().numpy()
            ref = O.three_interpolate_backward(g, idx, w, N2)
            tag = "interp B=%d C=%d N2=%d N1=%d hot=%d" % (B, C, N2, N1, hot)
        if not np.array_equal(got.view(np.uint32) & 0x7FFFFFFF | (got.view(np.uint32) & 0x80000000) * (got != 0),
                              ref.view(np.uint32) & 0x7FFFFFFF | (ref.view(np.uint32) & 0x80000000) * (ref != 0)):
            bad = np.argwhere(got != ref)
            print("MISMATCH case %d %s at %s: %r vs %r" % (case, tag, bad[0], got[tuple(bad[0])], ref[tuple(bad[0])]))
            sys.exit(1)
        if case % 50 == 0:
            print("ok  case %d %s" % (case, tag))
    print("%d cases, no mismatch" % case)


if __name__ == "__main__":
    main()
