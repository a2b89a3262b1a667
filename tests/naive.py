"""Second, independent restatement of the reference operators in plain numpy /
Python loops (small cases only).  Used to cross-check the C oracle: two
restatements written separately from the same .cu lines must agree bit for bit.
numpy float32 arithmetic rounds every operation, which is the canonical
(no-FMA) contract.  PN2U = reference pointnet2_utils.
"""
import numpy as np

F = np.float32


def dist2(p, c):
    """p (3,N) f32, c (3,) f32 -> (N,) f32;  PN2U/csrc/sampling_kernel.cu:84."""
    dx = p[0] - c[0]
    dy = p[1] - c[1]
    dz = p[2] - c[2]
    return (dx * dx + dy * dy) + dz * dz


def ref_block(n):
    cnt, x = 0, n - 1
    while x > 0:
        x >>= 1
        cnt += 1
    return max(16, min(1 << cnt, 512))


def fps_literal(points, m):
    """Thread-by-thread emulation of FarthestPointSampleKernel
    (PN2U/csrc/sampling_kernel.cu:49-119)."""
    B, _, N = points.shape
    bs = ref_block(N)
    out = np.zeros((B, m), dtype=np.int64)
    for b in range(B):
        p = points[b]
        temp = np.full(N, -1.0, dtype=F)
        cur = 0
        for i in range(1, m):
            d = dist2(p, p[:, cur])
            upd = (temp > d) | (temp < 0)
            temp = np.where(upd, d, temp)
            smem_d = np.zeros(bs, dtype=F)
            smem_i = np.full(bs, cur, dtype=np.int64)
            for t in range(min(bs, N)):
                js = np.arange(t, N, bs)
                vals = temp[js]
                best, besti = F(0), cur
                for j, v in zip(js, vals):
                    if v > best:
                        best, besti = v, j
                smem_d[t], smem_i[t] = best, besti
            off = bs // 2
            while off > 0:
                for t in range(off):
                    if smem_d[t] < smem_d[t + off]:
                        smem_d[t] = smem_d[t + off]
                        smem_i[t] = smem_i[t + off]
                off //= 2
            cur = int(smem_i[0])
            out[b, i] = cur
    return out


def ball_query(points, centroids, radius, K):
    """PN2U/csrc/ball_query_kernel.cu:33-76."""
    B, _, N = points.shape
    M = centroids.shape[2]
    r2 = F(radius) * F(radius)
    idx = np.zeros((B, M, K), dtype=np.int64)
    cnt = np.zeros((B, M), dtype=np.int64)
    for b in range(B):
        for m in range(M):
            d = dist2(points[b], centroids[b][:, m])
            hits = np.nonzero(d < r2)[0][:K]
            if len(hits):
                idx[b, m, :] = hits[0]
                idx[b, m, :len(hits)] = hits
            cnt[b, m] = len(hits)
    return idx, cnt


def three_nn(q, k):
    """PN2U/csrc/interpolate_kernel.cu:32-81 (ties keep the earlier key)."""
    B, _, N1 = q.shape
    N2 = k.shape[2]
    idx = np.zeros((B, N1, 3), dtype=np.int64)
    d2 = np.zeros((B, N1, 3), dtype=F)
    for b in range(B):
        for i in range(N1):
            d = dist2(k[b], q[b][:, i])
            order = np.lexsort((np.arange(N2), d))[:3]
            idx[b, i] = order
            d2[b, i] = d[order]
    return idx, d2


def group_points(points, index):
    B, C, N = points.shape
    return np.stack([points[b][:, index[b]] for b in range(B)], axis=0)


def gather_points(points, index):
    return np.stack([points[b][:, index[b]] for b in range(points.shape[0])], axis=0)


def interp_weights(d2, eps=1e-10):
    inv = F(1.0) / np.maximum(d2, F(eps))
    s = (inv[..., 0] + inv[..., 1]) + inv[..., 2]
    return inv / s[..., None]


def three_interpolate(feat, index, w):
    B, C, N2 = feat.shape
    out = np.zeros((B, C, index.shape[1]), dtype=F)
    for b in range(B):
        acc = np.zeros((C, index.shape[1]), dtype=F)
        for k in range(3):
            acc = acc + feat[b][:, index[b, :, k]] * w[b, :, k][None, :]
        out[b] = acc
    return out
