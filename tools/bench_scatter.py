#!/usr/bin/env python3
"""Timing of the two backward scatters at the network's training shapes: deterministic (sort + sequential sums) vs atomics.
Usage on a GPU box: python tools/bench_scatter.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import functions as F  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    print("| operator | shape | algorithmic MB | deterministic ms (GB/s) | atomic ms (GB/s) |\n|---|---|---:|---:|---:|")
    for name, B, C, N, M, K in (("group_points backward, SA1 xyz", 16, 3, 25600, 5120, 64),
                                ("group_points backward, SA2 features", 16, 256, 5120, 1024, 64),
                                ("group_points backward, SA3 features", 16, 512, 1024, 256, 64)):
        idx = torch.randint(0, N, (B, M, K), generator=g).to(dev)
        idx[:, :, 48:] = idx[:, :, :1]
        go = torch.randn(B, C, M, K, generator=g).to(dev)
        nbytes = B * (4 * C * M * K + 8 * M * K + 4 * C * N)
        row = []
        for mode in ("deterministic", "atomic"):
            F.set_backward_mode(mode)
            ms = timed(lambda: F._group_points_backward(go, idx, N))
            row.append("%.3f (%.0f)" % (ms, nbytes / ms / 1e6))
        print("| %s | B=%d C=%d N=%d M=%d K=%d | %.1f | %s | %s |" % (name, B, C, N, M, K, nbytes / 1e6, row[0], row[1]))
    for name, B, C, N2, N1 in (("three_interpolate backward, FP3", 16, 512, 5120, 25600),
                               ("three_interpolate backward, FP2", 16, 1024, 1024, 5120)):
        idx = torch.randint(0, N2, (B, N1, 3), generator=g).to(dev)
        w = torch.rand(B, N1, 3, generator=g).to(dev)
        go = torch.randn(B, C, N1, generator=g).to(dev)
        nbytes = B * (4 * C * N1 + 36 * N1 + 4 * C * N2)
        row = []
        for mode in ("deterministic", "atomic"):
            F.set_backward_mode(mode)
            ms = timed(lambda: F._interpolate_backward(go, idx, w, N2))
            row.append("%.3f (%.0f)" % (ms, nbytes / ms / 1e6))
        print("| %s | B=%d C=%d N2=%d N1=%d | %.1f | %s | %s |" % (name, B, C, N2, N1, nbytes / 1e6, row[0], row[1]))
    F.set_backward_mode("deterministic")


if __name__ == "__main__":
    main()
