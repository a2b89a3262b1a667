#!/usr/bin/env python3
"""the REFERENCE's Python network timed on this container's CPU cores.

IN-CONTAINER ONLY (needs /root/reference; nothing from it is copied or shipped).  Imports
`grasp_proposal.network_models.models.PointNet2_tcls.PointNet2` with the shipped config
(`configs/curvature_model.yaml:11-22`) and times `net({"scene_points": (1,3,25600)})` --
the span `grasp_proposal_test.py:72-76` times -- over two stand-ins for the CUDA-only
`pn2_ext` extension (the reference has no CPU path at all):

  naive   pure PyTorch operators written here: the "naive Python FPS / ball_query
          fallback" of BASELINE.json configs[0] (a Python loop of M FPS steps over torch
          vector ops; chunked distance matrices for ball query and 3-NN)
  oracle  the repo's C restatement of the .cu kernels (oracle/s4g_oracle.c, OpenMP)

on the reference's own sample scene (seeded 25 600-point subsample, the data fixture
tests/golden/pn2_real.npz) and on `tabletop-v1` scene 0.  Prints a markdown table for
BASELINE.md and checks that both stand-ins give the same indices (ties aside) and outputs.

    python tools/cpu_baseline_reference.py [--threads 8] [--reps 3]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference/inference"
EXT = "grasp_proposal.network_models.models.pointnet2_utils.pn2_ext"


def naive_pn2_ext():
    """Pure-PyTorch operators with the semantics of SURVEY Appendix A (exact ties aside:
    torch.argmax / topk pick the lowest index among equal values)."""
    m = types.ModuleType(EXT)

    def farthest_point_sample(points, num_centroids):
        B, _, N = points.shape
        out = torch.zeros((B, num_centroids), dtype=torch.int64)
        for b in range(B):
            p = points[b]                                   # (3, N)
            temp = torch.full((N,), float("inf"))
            cur = 0
            for i in range(1, num_centroids):
                d = ((p - p[:, cur:cur + 1]) ** 2).sum(dim=0)
                temp = torch.minimum(temp, d)
                cur = int(torch.argmax(temp))
                out[b, i] = cur
        return out

    def ball_query(points, centroids, radius, k):
        B, _, N = points.shape
        M = centroids.shape[2]
        r2 = torch.tensor(radius, dtype=torch.float32) ** 2
        idx = torch.zeros((B, M, k), dtype=torch.int64)
        cnt = torch.zeros((B, M), dtype=torch.int64)
        ar = torch.arange(N)
        for b in range(B):
            p = points[b].t().contiguous()                  # (N, 3)
            for m0 in range(0, M, 256):
                c = centroids[b][:, m0:m0 + 256].t()        # (m, 3)
                d = ((c[:, None, :] - p[None, :, :]) ** 2).sum(dim=2)
                hit = d < r2
                n = hit.sum(dim=1)
                order = torch.where(hit, ar[None, :], N).sort(dim=1)[0][:, :k]   # first k hits in index order
                first = order[:, :1]
                order = torch.where(order < N, order, first)                     # pad with the first hit
                order = torch.where(n[:, None] > 0, order, torch.zeros_like(order))
                idx[b, m0:m0 + 256] = order
                cnt[b, m0:m0 + 256] = n.clamp(max=k)
        return idx, cnt

    def group_points_forward(points, index):
        B, C, N = points.shape
        _, M, K = index.shape
        return points.unsqueeze(2).expand(B, C, M, N).gather(3, index.unsqueeze(1).expand(B, C, M, K))

    def point_search(query, key, k):
        B, _, N1 = query.shape
        idx = torch.zeros((B, N1, 3), dtype=torch.int64)
        dist = torch.zeros((B, N1, 3))
        for b in range(B):
            kk = key[b].t().contiguous()
            for n0 in range(0, N1, 1024):
                q = query[b][:, n0:n0 + 1024].t()
                d = ((q[:, None, :] - kk[None, :, :]) ** 2).sum(dim=2)
                v, i = torch.topk(d, 3, dim=1, largest=False, sorted=True)
                idx[b, n0:n0 + 1024] = i
                dist[b, n0:n0 + 1024] = v
        return idx, dist

    def interpolate_forward(feature, index, weight):
        B, C, N2 = feature.shape
        N1 = index.shape[1]
        g = feature.unsqueeze(2).expand(B, C, N1, N2).gather(3, index.unsqueeze(1).expand(B, C, N1, 3))
        return (g * weight.unsqueeze(1)).sum(dim=3)

    m.farthest_point_sample = farthest_point_sample
    m.ball_query = ball_query
    m.group_points_forward = group_points_forward
    m.point_search = point_search
    m.interpolate_forward = interpolate_forward
    m.group_points_backward = m.interpolate_backward = None
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    os.environ.setdefault("OMP_NUM_THREADS", str(args.threads))

    import gen_golden
    from s4g_release_amd import synth
    from s4g_release_amd.model import randomize_bn_
    oracle_ext = gen_golden.install_standin_pn2_ext()
    sys.path.insert(0, REF)
    from grasp_proposal.network_models.models.PointNet2_tcls import PointNet2 as RefPointNet2
    from grasp_proposal.network_models.models.pointnet2_utils import functions as ref_F
    naive_ext = naive_pn2_ext()

    torch.manual_seed(20260101)
    net = RefPointNet2(**gen_golden.FULL)
    randomize_bn_(net, 20260102)
    net.eval()

    real = np.load(os.path.join(ROOT, "tests", "golden", "pn2_real.npz"), allow_pickle=False)["points"]
    scenes = [("real scene 2638_view_0.p (seeded 25 600 of 48 902 points)", real),
              ("synthetic tabletop-v1 scene 0", synth.make_batch([0], 25600))]
    rows, outs = [], {}
    for label, pts in scenes:
        x = {"scene_points": torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32))}
        for name, ext in (("oracle (C, OpenMP)", oracle_ext), ("naive (pure PyTorch)", naive_ext)):
            ref_F.pn2_ext = ext
            ts = []
            reps = args.reps if "oracle" in name else 1
            for _ in range(reps):
                t0 = time.perf_counter()
                with torch.no_grad():
                    pred = net(x)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            med = ts[len(ts) // 2]
            outs[(label, name)] = {k: v.numpy() for k, v in pred.items()}
            rows.append((label, name, med, reps))
            print("%-60s %-22s %8.2f s  (%d run%s)" % (label, name, med, reps, "s" if reps > 1 else ""),
                  flush=True)
        a, b = outs[(label, "oracle (C, OpenMP)")], outs[(label, "naive (pure PyTorch)")]
        worst = max(float(np.max(np.abs(a[k] - b[k]))) for k in a)
        print("    max |oracle - naive| over the four outputs: %.3g" % worst, flush=True)
    print()
    print("| Input | `pn2_ext` stand-in | Forward time | Throughput | Hardware |")
    print("|---|---|---|---|---|")
    for label, name, med, reps in rows:
        print("| %s | %s | %.2f s (median of %d) | %.4f scenes/s | this container: %d x Xeon @ 2.1 GHz, "
              "torch %s CPU, %d threads |" % (label, name, med, reps, 1.0 / med, os.cpu_count(),
                                              torch.__version__, args.threads))


if __name__ == "__main__":
    main()
