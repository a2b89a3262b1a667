#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's
own Python network.

IN-CONTAINER ONLY: imports `/root/reference/inference/grasp_proposal/...`
(`PointNet2_tcls.PointNet2`, `pointnet2_utils/modules.py`, `nn_utils/*`) on CPU
with a stand-in for the CUDA extension `pn2_ext` (which cannot be built here)
backed by the C oracle.  What the fixtures pin is therefore the reference's
Python-level composition -- module wiring, channel bookkeeping, centroid
subtraction, concat orders, interpolation weights, conv/BN/ReLU semantics, head
order, state_dict key names -- around the oracle's operators.  Nothing from the
reference is copied into the repo: the fixtures are inputs and outputs only.

Outputs:
  tests/golden/pn2_small.npz   reduced config, full state_dict + all outputs
  tests/golden/pn2_full.npz    shipped config (curvature_model.yaml), weights by
                               seed + sha256, index hashes, outputs at 64 positions
  tests/golden/pn2_real.npz    the same for the reference's own sample scene
                               (inference/2638_view_0.p, `point_cloud` (3, 48 902) f32,
                               the input of grasp_proposal_test.py:17-32 / configs[0]):
                               a SEEDED 25 600-point subsample is stored as data (the
                               demo's np.random.choice is unseeded), shipped config,
                               weights by seed (no checkpoint ships with the reference)
  tests/golden/pn2_real_replace.npz   the same scene drawn WITH replacement (what the harness
                               feeds: exact copies among the 25 600 points)
S4G_GOLDEN_OUT=<dir> writes elsewhere (to add one fixture without rewriting the others).
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/inference"

from oracle import oracle as O  # noqa: E402
from s4g_release_amd import synth  # noqa: E402
from s4g_release_amd.model import randomize_bn_  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def install_standin_pn2_ext():
    """Register the oracle as `...pointnet2_utils.pn2_ext` (functions.py:2 imports it)."""
    name = "grasp_proposal.network_models.models.pointnet2_utils.pn2_ext"
    m = types.ModuleType(name)
    m.farthest_point_sample = lambda p, n: torch.from_numpy(O.fps(_np(p), int(n)))

    def ball_query(p, c, r, k):
        i, n = O.ball_query(_np(p), _np(c), float(r), int(k))
        return torch.from_numpy(i), torch.from_numpy(n)
    m.ball_query = ball_query
    m.group_points_forward = lambda p, i: torch.from_numpy(O.group_points(_np(p), _np(i)))
    m.group_points_backward = lambda g, i, n: torch.from_numpy(
        O.group_points_backward(_np(g), _np(i), int(n)))

    def point_search(q, k, n):
        assert int(n) == 3
        i, d = O.three_nn(_np(q), _np(k))
        return torch.from_numpy(i), torch.from_numpy(d)
    m.point_search = point_search
    m.interpolate_forward = lambda f, i, w: torch.from_numpy(
        O.three_interpolate(_np(f), _np(i), _np(w)))
    m.interpolate_backward = lambda g, i, w, n: torch.from_numpy(
        O.three_interpolate_backward(_np(g), _np(i), _np(w), int(n)))
    sys.modules[name] = m
    return m


def state_dict_sha256(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(_np(sd[k])).tobytes())
    return h.hexdigest()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


SMALL = dict(score_classes=3, num_centroids=(512, 128, 32), radius=(0.05, 0.12, 0.4),
             num_neighbours=(16, 16, 16),
             sa_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128)),
             fp_channels=((128, 128), (64, 64), (32, 32, 32)), num_fp_neighbours=(3, 3, 3),
             seg_channels=(64, 32, 32, 16), num_removal_directions=5, dropout_prob=0.5)
FULL = dict(score_classes=3, num_centroids=(5120, 1024, 256), radius=(0.02, 0.08, 0.32),
            num_neighbours=(64, 64, 64),
            sa_channels=((128, 128, 256), (256, 256, 512), (512, 512, 1024)),
            fp_channels=((1024, 1024), (512, 512), (256, 256, 256)), num_fp_neighbours=(3, 3, 3),
            seg_channels=(512, 256, 256, 128), num_removal_directions=5, dropout_prob=0.5)


def main():
    install_standin_pn2_ext()
    sys.path.insert(0, REF)
    from grasp_proposal.network_models.models.PointNet2_tcls import PointNet2 as RefPointNet2
    from grasp_proposal.network_models.models.pointnet2_utils import functions as ref_F
    out_dir = os.environ.get("S4G_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(8)

    # capture the index tensors the reference modules request
    captured = {}
    orig = {n: getattr(ref_F.pn2_ext, n) for n in ("farthest_point_sample", "ball_query",
                                                   "point_search")}

    def cap(name):
        def f(*a):
            r = orig[name](*a)
            captured.setdefault(name, []).append(r)
            return r
        return f
    for n in orig:
        setattr(ref_F.pn2_ext, n, cap(n))

    # ---- small config: everything stored
    seed = 1234
    torch.manual_seed(seed)
    net = RefPointNet2(**SMALL)
    randomize_bn_(net, seed + 1)
    net.eval()
    pts = synth.make_batch([11, 12], 2048)
    captured.clear()
    with torch.no_grad():
        pred = net({"scene_points": torch.from_numpy(pts)})
    sd = net.state_dict()
    blob = {"points": pts, "seed": np.int64(seed), "config_repr": np.array(repr(SMALL)),
            "state_dict_sha256": np.array(state_dict_sha256(sd))}
    for k, v in sd.items():
        blob["sd/" + k] = _np(v)
    for k, v in pred.items():
        blob["out/" + k] = _np(v)
    for li, r in enumerate(captured["farthest_point_sample"]):
        blob["fps%d" % li] = _np(r)
    for li, (i, c) in enumerate(captured["ball_query"]):
        blob["ball%d" % li] = _np(i).astype(np.int32)
        blob["cnt%d" % li] = _np(c).astype(np.int32)
    for li, (i, d) in enumerate(captured["point_search"]):
        blob["nn%d" % li] = _np(i).astype(np.int32)
        blob["nnd%d" % li] = _np(d)
    np.savez_compressed(os.path.join(out_dir, "pn2_small.npz"), **blob)
    print("pn2_small.npz: %d state_dict entries, outputs %s" % (
        len(sd), {k: tuple(v.shape) for k, v in pred.items()}))

    # ---- full (shipped) config: weights by seed, hashes + sampled outputs
    seed = 20260101
    torch.manual_seed(seed)
    net = RefPointNet2(**FULL)
    randomize_bn_(net, seed + 1)
    net.eval()
    sd = net.state_dict()
    assert len(sd) == 200, len(sd)
    pts = synth.make_batch([0], 25600)
    captured.clear()
    with torch.no_grad():
        pred = net({"scene_points": torch.from_numpy(pts)})
    pos = np.linspace(0, 25599, 64).astype(np.int64)
    blob = {"seed": np.int64(seed), "scene_id": np.int64(0), "points_sha256": np.array(sha(pts)),
            "state_dict_sha256": np.array(state_dict_sha256(sd)),
            "state_dict_keys": np.array(sorted(sd.keys())),
            "state_dict_shapes": np.array([repr(tuple(sd[k].shape)) for k in sorted(sd.keys())]),
            "num_params": np.int64(sum(p.numel() for p in net.parameters())),
            "positions": pos}
    for k, v in pred.items():
        blob["out/" + k] = _np(v)[:, :, pos]
        blob["outsum/" + k] = np.float64(_np(v).astype(np.float64).sum())
    for li, r in enumerate(captured["farthest_point_sample"]):
        a = _np(r)
        blob["fps%d_sha256" % li] = np.array(sha(a))
        blob["fps%d_head" % li] = a[:, :256]
    for li, (i, c) in enumerate(captured["ball_query"]):
        a, n = _np(i), _np(c)
        blob["ball%d_sha256" % li] = np.array(sha(a))
        blob["cnt%d_sha256" % li] = np.array(sha(n))
        blob["ball%d_head" % li] = a[:, :64].astype(np.int32)
        blob["ball%d_tail" % li] = a[:, -64:].astype(np.int32)
    for li, (i, d) in enumerate(captured["point_search"]):
        a, dd = _np(i), _np(d)
        blob["nn%d_sha256" % li] = np.array(sha(a))
        blob["nnd%d_sha256" % li] = np.array(sha(dd))
        blob["nn%d_head" % li] = a[:, :256].astype(np.int32)
    np.savez_compressed(os.path.join(out_dir, "pn2_full.npz"), **blob)
    print("pn2_full.npz: params %d, outputs at %d positions" % (blob["num_params"], len(pos)))

    # ---- the reference's sample scene through the same network
    import pickle
    with open(os.path.join(REF, "2638_view_0.p"), "rb") as f:
        scene = pickle.load(f)
    cloud = np.ascontiguousarray(scene["point_cloud"], dtype=np.float32)      # (3, 48902)
    pick = np.random.default_rng(2638).choice(cloud.shape[1], 25600, replace=False)
    pts = np.ascontiguousarray(cloud[:, pick][None])                         # (1, 3, 25600)
    captured.clear()
    with torch.no_grad():
        pred = net({"scene_points": torch.from_numpy(pts)})
    blob = {"seed": np.int64(seed), "points": pts, "source_points": np.int64(cloud.shape[1]),
            "subsample_seed": np.int64(2638), "state_dict_sha256": np.array(state_dict_sha256(sd)),
            "positions": pos}
    for k, v in pred.items():
        blob["out/" + k] = _np(v)[:, :, pos]
        blob["outsum/" + k] = np.float64(_np(v).astype(np.float64).sum())
    for li, r in enumerate(captured["farthest_point_sample"]):
        blob["fps%d_sha256" % li] = np.array(sha(_np(r)))
        blob["fps%d_head" % li] = _np(r)[:, :256]
    for li, (i, c) in enumerate(captured["ball_query"]):
        blob["ball%d_sha256" % li] = np.array(sha(_np(i)))
        blob["cnt%d_sha256" % li] = np.array(sha(_np(c)))
    for li, (i, d) in enumerate(captured["point_search"]):
        blob["nn%d_sha256" % li] = np.array(sha(_np(i)))
        blob["nnd%d_sha256" % li] = np.array(sha(_np(d)))
    np.savez_compressed(os.path.join(out_dir, "pn2_real.npz"), **blob)
    print("pn2_real.npz: %d of %d points of 2638_view_0.p" % (pts.shape[2], cloud.shape[1]))

    # ---- ... and drawn WITH replacement, which is what the harness itself does
    # (grasp_proposal_test.py:26-29: np.random.choice(..., replace=True) when the cloud is smaller than
    # NUM_INPUT, and the demo's own unseeded choice): ~21 % of the 25 600 points are exact copies
    pick = np.random.default_rng(26382).choice(cloud.shape[1], 25600, replace=True)
    pts = np.ascontiguousarray(cloud[:, pick][None])
    captured.clear()
    with torch.no_grad():
        pred = net({"scene_points": torch.from_numpy(pts)})
    blob = {"seed": np.int64(seed), "points": pts, "source_points": np.int64(cloud.shape[1]),
            "subsample_seed": np.int64(26382), "distinct_points": np.int64(len(np.unique(pick))),
            "state_dict_sha256": np.array(state_dict_sha256(sd)), "positions": pos}
    for k, v in pred.items():
        blob["out/" + k] = _np(v)[:, :, pos]
        blob["outsum/" + k] = np.float64(_np(v).astype(np.float64).sum())
    for li, r in enumerate(captured["farthest_point_sample"]):
        blob["fps%d_sha256" % li] = np.array(sha(_np(r)))
        blob["fps%d_head" % li] = _np(r)[:, :256]
    for li, (i, c) in enumerate(captured["ball_query"]):
        blob["ball%d_sha256" % li] = np.array(sha(_np(i)))
        blob["cnt%d_sha256" % li] = np.array(sha(_np(c)))
    for li, (i, d) in enumerate(captured["point_search"]):
        blob["nn%d_sha256" % li] = np.array(sha(_np(i)))
        blob["nnd%d_sha256" % li] = np.array(sha(_np(d)))
    np.savez_compressed(os.path.join(out_dir, "pn2_real_replace.npz"), **blob)
    print("pn2_real_replace.npz: %d draws, %d distinct points" % (pts.shape[2], blob["distinct_points"]))


if __name__ == "__main__":
    main()
