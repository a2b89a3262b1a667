"""Model-level stress with TRAINED-LIKE statistics (the checkpoint itself is not in the reference tree:
.MISSING_LARGE_BLOBS).  randomize_bn_'s gamma, sigma^2 in [0.5, 1.5] are benign for a contraction that scales
its operands into fp16's range by per-scene / per-tile powers of two; a trained network is not: BatchNorm gains
span decades and a few channels dominate.  Here every BatchNorm gets gamma log-uniform over 1e-3 .. 1e2 (three
outlier channels per layer 30 x larger), beta ~ gamma N(0, 0.5), and running statistics CALIBRATED on the
network's own activations (BN layers in train mode with momentum 1 for one pass), so that -- as after training --
sigma^2 and mu are what the preceding layers actually produce and span decades as well.  Full size, real scene
+ a synthetic one.  Yardstick: a float64 restatement (tests/ref64.py) on the same fp32 indices."""
import os

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
HEADS = ("score", "frame_R", "frame_t", "movable_logits")


def trained_like_(net, seed, calib):
    g = torch.Generator().manual_seed(seed)
    bns = [m for m in net.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
    with torch.no_grad():
        for m in bns:
            C = m.weight.numel()
            gamma = 10.0 ** (torch.rand(C, generator=g) * 5.0 - 3.0)
            gamma[torch.randperm(C, generator=g)[:3]] *= 30.0
            m.weight.copy_(gamma)
            m.bias.copy_(gamma * torch.randn(C, generator=g) * 0.5)
        net.eval()
        for m in bns:
            m.train()
            m.momentum = 1.0
        net(calib)                      # the reference-shaped modules on the HIP operators
        for m in bns:
            m.eval()
            m.momentum = 0.1
    return net


def _scenes():
    from s4g_release_amd import synth
    from tests import golden_util as GU
    real = GU.load("pn2_real.npz")["points"]          # 25 600-point subsample of the reference's 2638_view_0.p
    return {"real": np.ascontiguousarray(real[:1]), "tabletop": synth.make_batch([4], 25600)}


@pytest.mark.parametrize("scene", ["real", "tabletop"])
def test_trained_like_statistics_full_size(dev, scene):
    from oracle import pn2_forward
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls
    from tests.ref64 import forward64
    cfg = S4GConfig()
    pts = _scenes()[scene]
    torch.manual_seed(21)
    net = build_pointnet2_cls(cfg).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    trained_like_(net, 22, {"scene_points": d_pts})
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    gam = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("bn.weight")])
    var = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("bn.running_var")])
    assert gam.max() / gam.min() > 1e5 and var.max() / var.min() > 1e6      # decades, as asked
    ref = forward64(sd, pts, cfg.num_centroids, cfg.radius, cfg.num_neighbours)
    cpu32 = pn2_forward.forward(sd, pts, cfg.num_centroids, cfg.radius, cfg.num_neighbours)
    got = {p: FusedPointNet2(net, precision=p)({"scene_points": d_pts}) for p in ("f16x2", "bf16x3", "fp32")}
    report = {}
    for k in HEADS:
        scale = max(1.0, float(np.abs(ref[k]).max()))
        e = {p: float(np.abs(got[p][k].cpu().numpy().astype(np.float64) - ref[k]).max()) / scale for p in got}
        e["torch_cpu_fp32"] = float(np.abs(cpu32[k].astype(np.float64) - ref[k]).max()) / scale
        report[k] = (scale, e)
        print("%s[%s] max|ref| %.3g  err/scale: %s" % (scene, k, scale, {p: "%.2e" % v for p, v in e.items()}))
    for k, (scale, e) in report.items():
        assert np.isfinite(scale)
        # the north star's bar (1e-4 at the outputs' scale) and fp32-class: the split-fp16 contraction is no
        # further from the exact result than a few times what torch's own fp32 forward is
        assert e["f16x2"] < 1e-4 and e["bf16x3"] < 1e-4 and e["fp32"] < 1e-4, (k, e)
        assert e["f16x2"] < 8 * max(e["torch_cpu_fp32"], e["fp32"], 1e-7), (k, e)
