#!/usr/bin/env python3
"""Golden fixtures whose outputs DEPEND ON THE INPUT: the reference's own Python network with CALIBRATED
BatchNorm statistics.

IN-CONTAINER ONLY (imports /root/reference/inference/grasp_proposal/... exactly as tools/gen_golden.py does, the
CUDA extension replaced by the C oracle).  Why a second family: with seeded default init + `randomize_bn_`
(sigma^2 in [0.5, 1.5] on activations whose real variance is 1e-4 .. 1e-2) the signal dies in the first layers
and every output of pn2_{small,full,real}.npz is a per-channel constant (per-point std 4e-8) -- a permuted or
channel-reversed backbone passes 1e-4 against them.  Here the convolutions are seeded as before, gamma is
log-uniform over one decade, beta = gamma N(0, 0.5), and running_mean / running_var come from ONE train-mode pass
(momentum 1, BatchNorm layers only) THROUGH THE REFERENCE'S OWN MODULES (`s4g_release_amd.model.calibrate_bn_`
applied to the reference network), so every layer is re-normalised and the per-point spread of every output is of
the order of its magnitude.  The generator asserts that (std >= 0.05 x magnitude) so degeneracy cannot return.

  tests/golden/pn2_calib_small.npz  reduced config: points, whole state_dict, all outputs, all six level
                                    feature tensors, all index tensors
  tests/golden/pn2_calib_full.npz   shipped config: BatchNorm tensors by value (48 k floats; convolutions by
                                    seed + sha256), two scenes (tabletop-v1 scene 0, the reference's sample
                                    scene = pn2_real.npz's points): the four outputs at 256 positions + float64
                                    sums + per-channel spreads; through forward hooks on the reference's
                                    sa_modules[i] / fp_modules[i] a 64-position x 32-channel sample + float64 sum
                                    of every level's feature tensor; index hashes; and the measured distance
                                    torch-CPU-fp32 <-> float64 (tests/ref64.py) that the tolerance is set from
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/inference"

from s4g_release_amd import synth  # noqa: E402
from s4g_release_amd.model import calibrate_bn_  # noqa: E402
from tools.gen_golden import FULL, SMALL, install_standin_pn2_ext, sha, state_dict_sha256, _np  # noqa: E402

HEADS = ("score", "frame_R", "frame_t", "movable_logits")
DECADES = float(os.environ.get("S4G_CALIB_DECADES", "1.0"))
MIN_SPREAD = 0.05


def hook_levels(net, store):
    hs = []
    for i, m in enumerate(net.sa_modules):
        hs.append(m.register_forward_hook(lambda mod, a, out, i=i: store.__setitem__("sa%d" % i, out[1])))
    for i, m in enumerate(net.fp_modules):
        hs.append(m.register_forward_hook(lambda mod, a, out, i=i: store.__setitem__("fp%d" % i, out)))
    return hs


def capture_indices(ref_F, captured):
    orig = {n: getattr(ref_F.pn2_ext, n) for n in ("farthest_point_sample", "ball_query", "point_search")}

    def cap(name):
        def f(*a):
            r = orig[name](*a)
            captured.setdefault(name, []).append(r)
            return r
        return f
    for n in orig:
        setattr(ref_F.pn2_ext, n, cap(n))


def spread_check(name, a, strict=True):
    """a (B, C, N): the per-point spread of EVERY channel against that channel's own magnitude (a constant
    channel is what made the first fixture family blind), and the typical channel's against the tensor's."""
    a = a.astype(np.float64)
    std = a.std(axis=2)                           # (B, C)
    mag_c = np.abs(a).max(axis=2)                 # (B, C)
    live = mag_c > 0                              # a ReLU channel may be dead for a scene (all zeros): not degenerate wiring
    ratio = std[live] / mag_c[live]
    print("  %-22s channels %4d (dead %d)  std/|max| per channel: min %.3f median %.3f" % (
        name, a.shape[1], int((~live).sum()), ratio.min(), np.median(ratio)))
    if strict:
        assert ratio.min() >= MIN_SPREAD, (name, float(ratio.min()))
    assert np.median(ratio) >= MIN_SPREAD and (~live).mean() < 0.1, (name, float(np.median(ratio)))
    return std


_CTX = {}


def setup():
    """Import the reference network (once) over the oracle stand-in; returns (RefPointNet2, captured index dict)."""
    if not _CTX:
        install_standin_pn2_ext()
        sys.path.insert(0, REF)
        from grasp_proposal.network_models.models.PointNet2_tcls import PointNet2 as RefPointNet2
        from grasp_proposal.network_models.models.pointnet2_utils import functions as ref_F
        torch.set_num_threads(8)
        captured = {}
        capture_indices(ref_F, captured)
        _CTX.update(net=RefPointNet2, captured=captured)
    return _CTX["net"], _CTX["captured"]


def gen_small(out_dir):
    """tests/golden/pn2_calib_small.npz (a few seconds; tests/test_reference_dropin.py regenerates it and compares)."""
    RefPointNet2, captured = setup()
    from tests.ref64 import forward64
    # ---------------- reduced config: everything stored
    seed = 4321
    torch.manual_seed(seed)
    net = RefPointNet2(**SMALL)
    pts = synth.make_batch([21, 22], 2048)
    t_pts = torch.from_numpy(pts)
    calibrate_bn_(net, seed + 1, {"scene_points": t_pts}, decades=DECADES)
    assert not net.training and not any(m.training for m in net.modules())
    feats = {}
    hook_levels(net, feats)
    captured.clear()
    with torch.no_grad():
        pred = net({"scene_points": t_pts})
    sd = net.state_dict()
    blob = {"points": pts, "seed": np.int64(seed), "decades": np.float64(DECADES),
            "config_repr": np.array(repr(SMALL)), "state_dict_sha256": np.array(state_dict_sha256(sd))}
    for k, v in sd.items():
        blob["sd/" + k] = _np(v)
    for k in HEADS:
        blob["out/" + k] = _np(pred[k])
        spread_check("small/" + k, blob["out/" + k])
    for k, v in feats.items():
        blob["feat/" + k] = _np(v)
        spread_check("small/feat/" + k, blob["feat/" + k], strict=False)
    for li, r in enumerate(captured["farthest_point_sample"]):
        blob["fps%d" % li] = _np(r)
    for li, (i, c) in enumerate(captured["ball_query"]):
        blob["ball%d" % li] = _np(i).astype(np.int32)
        blob["cnt%d" % li] = _np(c).astype(np.int32)
    for li, (i, d) in enumerate(captured["point_search"]):
        blob["nn%d" % li] = _np(i).astype(np.int32)
        blob["nnd%d" % li] = _np(d)
    for b in range(pts.shape[0]):                   # float64 arithmetic on the same weights and indices: the yardstick
        ref = forward64(sd, pts[b:b + 1], SMALL["num_centroids"], SMALL["radius"], SMALL["num_neighbours"])
        for k in HEADS:
            blob.setdefault("out64/" + k, np.zeros(blob["out/" + k].shape, np.float64))[b] = ref[k][0]
    for k in HEADS:
        scale = max(1.0, float(np.abs(blob["out64/" + k]).max()))
        blob["margin/" + k] = np.array([np.abs(blob["out/" + k] - blob["out64/" + k]).max() / scale, scale])
        print("  small %-14s torch-fp32 vs float64: %.2e of scale %.3f" % (k, blob["margin/" + k][0], scale))
    np.savez_compressed(os.path.join(out_dir, "pn2_calib_small.npz"), **blob)
    print("pn2_calib_small.npz:", {k: "std/|max| %.2f" % (blob["out/" + k].std(axis=2).mean() / np.abs(blob["out/" + k]).max())
                                   for k in HEADS})


def gen_full(out_dir):
    """tests/golden/pn2_calib_full.npz (about two minutes)."""
    RefPointNet2, captured = setup()
    from tests.ref64 import forward64
    # ---------------- shipped config, two scenes
    seed = 20260606
    torch.manual_seed(seed)
    net = RefPointNet2(**FULL)
    real = np.load(os.path.join(ROOT, "tests", "golden", "pn2_real.npz"))["points"]      # (1, 3, 25600)
    scenes = {"tabletop": synth.make_batch([0], 25600), "real": np.ascontiguousarray(real[:1])}
    names = list(scenes)
    pts = np.concatenate([scenes[n] for n in names], axis=0)
    t_pts = torch.from_numpy(pts)
    calibrate_bn_(net, seed + 1, {"scene_points": t_pts}, decades=DECADES)       # BOTH scenes calibrate
    sd = net.state_dict()
    assert len(sd) == 200
    feats = {}
    hook_levels(net, feats)
    captured.clear()
    with torch.no_grad():
        pred = net({"scene_points": t_pts})
    rng = np.random.default_rng(606)
    pos = np.sort(rng.choice(25600, 256, replace=False)).astype(np.int64)
    blob = {"seed": np.int64(seed), "decades": np.float64(DECADES), "scenes": np.array(names),
            "state_dict_sha256": np.array(state_dict_sha256(sd)), "positions": pos,
            "points_sha256": np.array([sha(scenes[n]) for n in names])}
    for k, v in sd.items():
        if ".bn." in k:
            blob["bn/" + k] = _np(v)
    for k in HEADS:
        a = _np(pred[k])
        std = spread_check("full/" + k, a)
        for s, n in enumerate(names):
            blob["out/%s/%s" % (n, k)] = a[s][:, pos]
            blob["outsum/%s/%s" % (n, k)] = np.float64(a[s].astype(np.float64).sum())
            blob["outstd/%s/%s" % (n, k)] = std[s]
    for lv, v in feats.items():
        a = _np(v)                                   # (2, C, n)
        spread_check("full/feat/" + lv, a, strict=False)
        C, n = a.shape[1:]
        fpos = np.sort(rng.choice(n, 64, replace=False)).astype(np.int64)
        fch = np.sort(rng.choice(C, 32, replace=False)).astype(np.int64)
        blob["featpos/" + lv], blob["featch/" + lv] = fpos, fch
        for s, nme in enumerate(names):
            blob["feat/%s/%s" % (nme, lv)] = a[s][np.ix_(fch, fpos)]
            blob["featsum/%s/%s" % (nme, lv)] = np.float64(a[s].astype(np.float64).sum())
            blob["featabs/%s/%s" % (nme, lv)] = np.float64(np.abs(a[s]).max())
    for s, n in enumerate(names):                 # per scene, so that a B = 1 run can be checked
        for li, r in enumerate(captured["farthest_point_sample"]):
            blob["fps%d_sha256/%s" % (li, n)] = np.array(sha(_np(r)[s:s + 1]))
        for li, (i, c) in enumerate(captured["ball_query"]):
            blob["ball%d_sha256/%s" % (li, n)] = np.array(sha(_np(i)[s:s + 1]))
            blob["cnt%d_sha256/%s" % (li, n)] = np.array(sha(_np(c)[s:s + 1]))
        for li, (i, d) in enumerate(captured["point_search"]):
            blob["nn%d_sha256/%s" % (li, n)] = np.array(sha(_np(i)[s:s + 1]))
            blob["nnd%d_sha256/%s" % (li, n)] = np.array(sha(_np(d)[s:s + 1]))
    # the yardstick: how far torch's CPU fp32 kernels under the reference network sit from float64 arithmetic
    for s, n in enumerate(names):
        ref = forward64(sd, scenes[n], FULL["num_centroids"], FULL["radius"], FULL["num_neighbours"])
        for k in HEADS:
            scale = max(1.0, float(np.abs(ref[k]).max()))
            e = float(np.abs(_np(pred[k])[s:s + 1].astype(np.float64) - ref[k]).max()) / scale
            blob["margin/%s/%s" % (n, k)] = np.array([e, scale])
            blob["out64/%s/%s" % (n, k)] = ref[k][0][:, pos]
            print("%-8s %-14s max|ref| %8.3f  per-point std (median over channels) %7.3f  torch-fp32 vs float64: %.2e of scale"
                  % (n, k, np.abs(ref[k]).max(), float(np.median(blob["outstd/%s/%s" % (n, k)])), e))
    np.savez_compressed(os.path.join(out_dir, "pn2_calib_full.npz"), **blob)
    print("pn2_calib_full.npz: %d BatchNorm tensors by value, %d positions, levels %s" % (
        sum(1 for k in blob if k.startswith("bn/")), len(pos), sorted(feats)))


def main():
    out_dir = os.environ.get("S4G_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
    gen_small(out_dir)
    gen_full(out_dir)


if __name__ == "__main__":
    main()
