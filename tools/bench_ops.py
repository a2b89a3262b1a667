#!/usr/bin/env python3
"""Operator-level timing on the GPU box (HIP events on the launch stream).
Usage: python tools/bench_ops.py [--batch 16] [--ops fps,ball,group,nn,interp]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import functions as F, synth  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--ops", default="fps,ball,group,nn,interp")
    ap.add_argument("--variant", default="tabletop-v1")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    pts = torch.from_numpy(synth.make_batch(list(range(B)), 25600, variant=a.variant)).to(dev)
    ops = a.ops.split(",")
    idx1 = F.farthest_point_sample(pts, 5120)
    c1 = F.gather_points(pts, idx1)
    idx2 = F.farthest_point_sample(c1, 1024)
    c2 = F.gather_points(c1, idx2)
    if "fps" in ops:
        for name, x, m in (("fps 25600->5120", pts, 5120), ("fps 5120->1024", c1, 1024),
                           ("fps 1024->256", c2, 256)):
            ms = timeit(lambda: F.farthest_point_sample(x, m), reps=3, warm=1)
            print("%-28s B=%d  %9.3f ms   %.3f us/step" % (name, B, ms, 1e3 * ms / (m - 1)))
    if "fps51k" in ops:   # configs[4] cloud size: the hybrid kernel (x + min-distance in registers)
        big = torch.from_numpy(synth.make_batch(list(range(B)), 51200, variant=a.variant)).to(dev)
        ms = timeit(lambda: F.farthest_point_sample(big, 5120), reps=2, warm=1)
        print("%-28s B=%d  %9.3f ms   %.3f us/step" % ("fps 51200->5120", B, ms, 1e3 * ms / 5119))
    if "ball" in ops or "ball1" in ops:
        sizes = (("ball 25600/5120 r.02", pts, c1, 0.02), ("ball 5120/1024 r.08", c1, c2, 0.08))
        for name, x, c, r in (sizes if "ball" in ops else sizes[:1]):
            ms = timeit(lambda: F.ball_query(x, c, r, 64))
            N, M = x.shape[2], c.shape[2]
            nb = B * (12 * N + 12 * M + 8 * M * 64 + 8 * M)
            print("%-28s B=%d  %9.3f ms   %.1f GB/s alg" % (name, B, ms, nb / ms / 1e6))
    if "qgroup" in ops:
        ms = timeit(lambda: F.query_and_group(pts, c1, 0.02, 64))
        nb = B * (12 * 25600 + 12 * 5120 + 8 * 5120 * 64 + 8 * 5120) + \
            B * (4 * 3 * 25600 + 8 * 5120 * 64 + 4 * 3 * 5120 * 64)
        print("%-28s B=%d  %9.3f ms   %.1f GB/s alg" % ("query+group SA1 one pass", B, ms, nb / ms / 1e6))
    if "group" in ops:
        gi, _ = F.ball_query(pts, c1, 0.02, 64)
        ms = timeit(lambda: F.group_points(pts, gi))
        nb = B * (4 * 3 * 25600 + 8 * 5120 * 64 + 4 * 3 * 5120 * 64)
        print("%-28s B=%d  %9.3f ms   %.1f GB/s alg" % ("group xyz SA1", B, ms, nb / ms / 1e6))
    if "groupfeat" in ops:   # feature grouping of the reference-shaped modules path (SA2 / SA3 sizes)
        for name, C, N, M, r, src, ctr in (("group feat SA2 C=256", 256, 5120, 1024, 0.08, c1, c2),):
            gi, _ = F.ball_query(src, ctr, r, 64)
            feat = torch.randn(B, C, N, device=dev)
            ms = timeit(lambda: F.group_points(feat, gi))
            nb = B * (4 * C * N + 8 * M * 64 + 4 * C * M * 64)
            print("%-28s B=%d  %9.3f ms   %.1f GB/s alg" % (name, B, ms, nb / ms / 1e6))
    if "nn" in ops:
        for name, q, k in (("3nn 25600<-5120", pts, c1), ("3nn 5120<-1024", c1, c2)):
            ms = timeit(lambda: F.search_nn_distance(q, k, 3))
            print("%-28s B=%d  %9.3f ms" % (name, B, ms))
        ms = timeit(lambda: F.three_nn_weights_grid(pts, c1, 0.02))
        nb = B * (12 * 5120 + 12 * 25600 + 24 * 25600 + 12 * 25600)
        print("%-28s B=%d  %9.3f ms   %.1f GB/s alg" % ("3nn grid 25600<-5120", B, ms, nb / ms / 1e6))
    if "interp" in ops:
        feat = torch.randn(B, 512, 5120, device=dev)
        i3, d3 = F.search_nn_distance(pts, c1, 3)
        w = F.interp_weights(d3)
        ms = timeit(lambda: F.feature_interpolate(feat, i3, w))
        nb = B * (4 * 512 * 5120 + 36 * 25600 + 4 * 512 * 25600)
        print("%-28s B=%d  %9.3f ms   %.1f GB/s alg" % ("interp FP3 C=512", B, ms, nb / ms / 1e6))


if __name__ == "__main__":
    main()
