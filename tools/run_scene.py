#!/usr/bin/env python3
"""One scene through the MI355X fast path: the counterpart of the reference's demo harness
`grasp_proposal/grasp_proposal_test.py:17-86` (load a cloud -> subsample to 25 600 points ->
(1, 3, N) fp32 -> forward -> time -> dump), without its open3d / yacs dependencies.

    python tools/run_scene.py CLOUD [--points 25600] [--seed 2638] [--weights CKPT.pth]
                              [--reps 20] [--out predictions.npz] [--topk 50]

CLOUD     .p / .pkl (the reference's scene pickle: dict with `point_cloud` (3, N) f32, as
          `inference/2638_view_0.p`), .npy / .npz (array or `points` / `point_cloud` entry,
          (3, N), (N, 3) or (1, 3, N)), or `synthetic:<scene_id>` (bench.py's tabletop-v1).
Subsample the reference draws `np.random.choice(N, 25600, replace = N < 25600)` UNSEEDED
          (`grasp_proposal_test.py:26-29`); here the draw is seeded (default 2638, the seed the
          golden fixture `tests/golden/pn2_real.npz` was made with) so runs are repeatable.
Weights   `--weights` takes the reference's checkpoint format ({"model": state_dict}, optional
          `module.` prefixes; `utils/checkpoint.py:31,54-55,81-88`).  The pretrained files do not
          ship with the reference, so the default is the seeded random model the fixtures use.
Timing    the reference times `model(data_batch)` without a device sync (`:72-78`); here each
          repetition is bracketed by synchronisation, median / p10 / p90 are reported and the
          host -> device copy of the cloud is timed apart.
Needs a HIP device and libs4g_hip.so (no CPU fallback).
"""
import argparse
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_cloud(path):
    """-> (3, N) float32"""
    if path.startswith("synthetic:"):
        from s4g_release_amd import synth
        return synth.make_batch([int(path.split(":", 1)[1])], 48902)[0]
    if path.endswith((".p", ".pkl", ".pickle")):
        with open(path, "rb") as f:
            obj = pickle.load(f)
        a = obj["point_cloud"] if isinstance(obj, dict) else obj
    elif path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        a = z["points"] if "points" in z.files else z["point_cloud"]
    else:
        a = np.load(path, allow_pickle=False)
    a = np.asarray(a, dtype=np.float32)
    if a.ndim == 3:
        a = a[0]
    if a.shape[0] != 3 and a.shape[1] == 3:
        a = a.T
    if a.ndim != 2 or a.shape[0] != 3:
        raise ValueError("cloud must be (3, N), (N, 3) or (1, 3, N); got %s" % (a.shape,))
    return np.ascontiguousarray(a)


def subsample(cloud, n, seed):
    """Seeded version of grasp_proposal_test.py:26-29 (same call the golden fixture used)."""
    total = cloud.shape[1]
    if total == n:
        return cloud
    pick = np.random.default_rng(seed).choice(total, n, replace=total < n)
    return np.ascontiguousarray(cloud[:, pick])


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("cloud")
    ap.add_argument("--points", type=int, default=25600)
    ap.add_argument("--seed", type=int, default=2638)
    ap.add_argument("--weights", default="calibrated",
                    help="a reference-format checkpoint (.pth: torch.save({'model': state_dict})), 'calibrated' (default: the "
                         "golden run's network of tests/golden/pn2_calib_full.npz -- seeded convolutions, BatchNorm statistics "
                         "calibrated through the reference's own modules: outputs that depend on the input; no trained checkpoint "
                         "ships with the reference) or 'seeded' (default init + randomize_bn_: per-channel-constant outputs)")
    ap.add_argument("--model-seed", type=int, default=20260101)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--precision", default=None)
    ap.add_argument("--out", default=None, help="write the four head tensors (+ top-K poses) as .npz")
    ap.add_argument("--topk", type=int, default=0, help="also decode the K best grasp frames on device")
    args = ap.parse_args()

    import torch
    from s4g_release_amd import _cabi
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, load_checkpoint, randomize_bn_
    assert torch.cuda.is_available(), "run_scene.py needs a HIP device"
    _cabi.lib()
    dev = torch.device("cuda:0")

    cloud = load_cloud(args.cloud)
    pts = subsample(cloud, args.points, args.seed)[None]                      # (1, 3, N)
    if args.weights == "calibrated":
        sys.path.insert(0, ROOT)
        from tests import golden_util as GU      # data fixture (BatchNorm tensors by value, sha-checked), not the oracle
        net = GU.calib_full_model()
    else:
        torch.manual_seed(args.model_seed)
        net = build_pointnet2_cls(S4GConfig())
        if args.weights == "seeded":
            randomize_bn_(net, args.model_seed + 1)
        else:
            load_checkpoint(net, args.weights)
    net = net.to(dev).eval()
    run = FusedPointNet2(net, precision=args.precision)

    host = torch.from_numpy(pts).pin_memory()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    x = host.to(dev, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    h2d_ms = e0.elapsed_time(e1)
    batch = {"scene_points": x}
    with torch.no_grad():
        for _ in range(3):
            pred = run(batch)
        ts = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred = run(batch)                      # the span grasp_proposal_test.py:72-76 times
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    q = lambda f: ts[min(len(ts) - 1, int(round(f * (len(ts) - 1))))]
    report = {"cloud": args.cloud, "source_points": int(cloud.shape[1]), "points": int(pts.shape[2]),
              "subsample_seed": args.seed, "weights": args.weights if args.weights != "seeded" else "seeded random (%d)" % args.model_seed,
              "precision": run.precision, "forward_ms": {"median": round(q(0.5), 3), "p10": round(q(0.1), 3),
                                                         "p90": round(q(0.9), 3), "reps": args.reps},
              "scenes_per_sec": round(1e3 / q(0.5), 2), "h2d_ms": round(h2d_ms, 4),
              "outputs": {k: list(v.shape) for k, v in pred.items()}}
    blob = {k: v.cpu().numpy() for k, v in pred.items()}
    if args.topk > 0:
        from s4g_release_amd import postprocess
        H, score, index = postprocess.decode_top_poses(pred, x, args.topk)
        torch.cuda.synchronize()
        blob.update(pose_H=H.cpu().numpy(), pose_score=score.cpu().numpy(), pose_index=index.cpu().numpy())
        report["topk"] = {"k": args.topk, "best_score": float(score.max())}
    if args.out:
        np.savez_compressed(args.out, points=pts, **blob)
        report["out"] = args.out
    print(json.dumps(report))


if __name__ == "__main__":
    main()
