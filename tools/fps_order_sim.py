#!/usr/bin/env python3
"""CPU simulation of the pruned FPS kernels' bookkeeping: how many 64-point groups a new centroid
can still change ("touched groups per pick"), as a function of the ORDER the points are grouped
in.  The kernels are exact for any permutation; the order only sets this number.  numpy, fp64,
same rule as the kernel: a group is touched iff the squared distance from the centroid to the
group's box is below the group's largest running min-distance.

Usage: python tools/fps_order_sim.py [variant] [N] [M]      (default tabletop-v1 25600 5120)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import synth  # noqa: E402


def iso_morton(P, bits):
    """round 2: `bits` per axis on box-normalised coordinates, z-y-x interleave"""
    lo = P.min(0)
    ext = P.max(0) - lo
    q = np.minimum(((P - lo) / ext * ((1 << bits) - 1)).astype(np.int64), (1 << bits) - 1)
    key = np.zeros(len(P), np.int64)
    for i in range(bits - 1, -1, -1):
        for a in (2, 1, 0):
            key = (key << 1) | ((q[:, a] >> i) & 1)
    return np.argsort(key, kind="stable")


def dealt_morton(P, total):
    """`total` bits dealt to the axes by extent: the next bit halves the axis whose cells are longest"""
    lo = P.min(0)
    ext = P.max(0) - lo
    b, seq = [0, 0, 0], []
    for _ in range(total):
        a = int(np.argmax([ext[i] / (1 << b[i]) for i in range(3)]))
        seq.append(a)
        b[a] += 1
    q = [np.minimum(((P[:, i] - lo[i]) / ext[i] * (1 << b[i])).astype(np.int64), (1 << b[i]) - 1) for i in range(3)]
    used, key = [0, 0, 0], np.zeros(len(P), np.int64)
    for a in seq:
        used[a] += 1
        key = (key << 1) | ((q[a] >> (b[a] - used[a])) & 1)
    return np.argsort(key, kind="stable"), b


def hilbert_index(x, y, bits):
    x, y, d = x.copy(), y.copy(), np.zeros_like(x)
    n = 1 << bits
    s = n >> 1
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, n - 1 - x, x)
        y = np.where(flip, n - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


def hilbert_2d(P, nb, nz):
    """2-D Hilbert curve over the two long axes (square cells), the short axis as nz minor bits"""
    lo = P.min(0)
    ext = P.max(0) - lo
    a, b, c = np.argsort(-ext)
    e = max(ext[a], ext[b])
    qa = np.minimum(((P[:, a] - lo[a]) / e * (1 << nb)).astype(np.int64), (1 << nb) - 1)
    qb = np.minimum(((P[:, b] - lo[b]) / e * (1 << nb)).astype(np.int64), (1 << nb) - 1)
    qc = np.minimum(((P[:, c] - lo[c]) / max(ext[c], 1e-30) * (1 << nz)).astype(np.int64), (1 << nz) - 1) \
        if nz > 0 else np.zeros(len(P), np.int64)
    return np.argsort((hilbert_index(qa, qb, nb) << nz) | qc, kind="stable")


def touched_per_pick(P, perm, M, dense=48):
    Q = P[perm]
    N = len(Q)
    G = N // 64
    gl = Q[:G * 64].reshape(G, 64, 3).min(1)
    gh = Q[:G * 64].reshape(G, 64, 3).max(1)
    d = np.full(N, np.inf)
    cur = int(np.where(perm == 0)[0][0])
    touched = steps = 0
    for i in range(1, M):
        c = Q[cur]
        if i <= dense:
            d = np.minimum(d, ((Q - c) ** 2).sum(1))
        else:
            gm = d[:G * 64].reshape(G, 64).max(1)
            box = np.maximum(np.maximum(gl - c, c - gh), 0)
            t = np.where((box ** 2).sum(1) < gm)[0]
            touched += len(t)
            steps += 1
            idx = (t[:, None] * 64 + np.arange(64)).ravel()
            d[idx] = np.minimum(d[idx], ((Q[idx] - c) ** 2).sum(1))
        cur = int(d.argmax())
    return touched / steps


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "tabletop-v1"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 25600
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 5120
    P = synth.make_scene(0, N, variant=variant).T.astype(np.float64)
    print("%s, N = %d, M = %d, extents %s" % (variant, N, M, np.round(P.max(0) - P.min(0), 3)))
    print("| order | touched groups per pick |\n|---|---:|")
    print("| Morton, 10 bits per axis on box-normalised coordinates (round 2) | %.2f |" % touched_per_pick(P, iso_morton(P, 10), M), flush=True)
    print("| Morton, 5 bits per axis | %.2f |" % touched_per_pick(P, iso_morton(P, 5), M), flush=True)
    perm, b = dealt_morton(P, 15)
    print("| 15 Morton bits dealt by extent %s | %.2f |" % (b, touched_per_pick(P, perm, M)), flush=True)
    for nb, nz in ((6, 3), (7, 1)):
        print("| 2-D Hilbert %d x %d over the long axes + %d minor bits | %.2f |" % (
            1 << nb, 1 << nb, nz, touched_per_pick(P, hilbert_2d(P, nb, nz), M)), flush=True)


if __name__ == "__main__":
    main()
