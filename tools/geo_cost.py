"""What the coordinate-only work of the NEXT batches costs the pipelined step: the bench loop with
selected geometry operators answered from a cache (measurement only -- results of the cached runs
are the same tensors, the operators simply do not run).

  python tools/geo_cost.py [--steps 30] [--batch 16]

Prints scenes/s and ms/step per ablation, interleaved over --rounds rounds."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import synth                                  # noqa: E402
from s4g_release_amd.fused import FusedPointNet2                   # noqa: E402
from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_   # noqa: E402

OPS = ("_fps_gather", "_fps_prefix_check", "_ball_query", "_group_rel_xyz_unique", "_group_rel_xyz", "_three_nn")
CASES = [("everything runs", ()),
         ("no FPS (all levels + prefix check)", ("_fps_gather", "_fps_prefix_check")),
         ("no ball queries", ("_ball_query",)),
         ("no row gathers", ("_group_rel_xyz_unique", "_group_rel_xyz")),
         ("no 3-NN", ("_three_nn",)),
         ("no geometry at all", OPS)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=25600)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--precision", default="f16x2")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(20260101)
    net = build_pointnet2_cls(S4GConfig())
    randomize_bn_(net, 20260101)
    net = net.to(dev).eval()
    run = FusedPointNet2(net, precision=args.precision)
    data = {"scene_points": torch.from_numpy(synth.make_batch(range(args.batch), args.points)).to(dev)}
    orig = {n: getattr(run, n) for n in OPS}
    cache = {}

    def cached(name):
        def f(*a, **kw):
            key = (name,) + tuple(tuple(t.shape) if isinstance(t, torch.Tensor) else t for t in a) + \
                tuple(sorted((k, v is not None) for k, v in kw.items()))
            if key not in cache:
                cache[key] = orig[name](*a, **kw)
            return cache[key]
        return f

    def loop(n):
        pending = []
        for _ in range(n):
            pending.append(run.submit(data))
            if len(pending) > 2:
                pending.pop(0).result()
        while pending:
            pending.pop(0).result()

    with torch.no_grad():
        for r in range(args.rounds):
            for label, off in CASES:
                for n in OPS:
                    setattr(run, n, cached(n) if n in off else orig[n])
                loop(args.warmup)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop(args.steps)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / args.steps
                print("round %d  %-38s %7.3f ms/step  %7.1f scenes/s" % (r, label, ms, args.batch * 1e3 / ms), flush=True)


if __name__ == "__main__":
    main()
