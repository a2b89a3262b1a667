"""GPU parity of the whole forward pass against the golden fixtures captured
from the reference's Python network and against the CPU oracle forward.
Tolerance: 1e-4 absolute on every regressed output (BASELINE.json north_star);
sampling / grouping / 3-NN indices bit-exact."""
import numpy as np
import pytest
import torch

from tests import golden_util as GU

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _capture_indices(F):
    """Wrap the raw entry points to record the index tensors a forward produces."""
    rec = {"fps": [], "ball": [], "nn": []}
    orig = (F._farthest_point_sample, F._ball_query, F._point_search, F.query_and_group)

    def fps(*a):
        r = orig[0](*a); rec["fps"].append(r); return r

    def ball(*a):
        r = orig[1](*a); rec["ball"].append(r); return r

    def nn(*a):
        r = orig[2](*a); rec["nn"].append(r); return r
    def qgroup(*a):      # QueryGrouper's one-pass form: (index, count, grouped xyz)
        r = orig[3](*a); rec["ball"].append((r[0], r[1])); return r
    F._farthest_point_sample, F._ball_query, F._point_search, F.query_and_group = fps, ball, nn, qgroup
    return rec, orig


def _restore(F, orig):
    F._farthest_point_sample, F._ball_query, F._point_search, F.query_and_group = orig


def test_reference_shaped_model_small_golden(dev):
    from s4g_release_amd import functions as F
    from s4g_release_amd.model import PointNet2
    g = GU.load("pn2_small.npz")
    cfg = GU.small_config(g)
    net = PointNet2(**cfg)
    net.load_state_dict(GU.small_state_dict(g), strict=True)
    net = net.to(dev).eval()
    rec, orig = _capture_indices(F)
    try:
        with torch.no_grad():
            pred = net({"scene_points": torch.from_numpy(g["points"]).to(dev)})
    finally:
        _restore(F, orig)
    for li in range(3):
        assert np.array_equal(rec["fps"][li].cpu().numpy(), g["fps%d" % li])
        assert np.array_equal(rec["ball"][li][0].cpu().numpy(), g["ball%d" % li])
        assert np.array_equal(rec["ball"][li][1].cpu().numpy(), g["cnt%d" % li])
        assert np.array_equal(rec["nn"][li][0].cpu().numpy(), g["nn%d" % li])
        assert np.array_equal(rec["nn"][li][1].cpu().numpy(), g["nnd%d" % li])
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        err = np.max(np.abs(pred[k].cpu().numpy() - g["out/" + k]))
        assert err < TOL, (k, err)


def test_reference_shaped_model_full_golden(dev):
    from s4g_release_amd import functions as F, synth
    g = GU.load("pn2_full.npz")
    net = GU.build_full_model(int(g["seed"]))
    assert GU.state_dict_sha256(net.state_dict()) == str(g["state_dict_sha256"])
    net = net.to(dev)
    pts = synth.make_batch([int(g["scene_id"])], 25600)
    assert GU.sha(pts) == str(g["points_sha256"])
    rec, orig = _capture_indices(F)
    try:
        with torch.no_grad():
            pred = net({"scene_points": torch.from_numpy(pts).to(dev)})
    finally:
        _restore(F, orig)
    for li in range(3):
        assert GU.sha(rec["fps"][li].cpu().numpy()) == str(g["fps%d_sha256" % li])
        assert GU.sha(rec["ball"][li][0].cpu().numpy()) == str(g["ball%d_sha256" % li])
        assert GU.sha(rec["ball"][li][1].cpu().numpy()) == str(g["cnt%d_sha256" % li])
        assert GU.sha(rec["nn"][li][0].cpu().numpy()) == str(g["nn%d_sha256" % li])
        assert GU.sha(rec["nn"][li][1].cpu().numpy()) == str(g["nnd%d_sha256" % li])
    pos = torch.from_numpy(g["positions"]).to(dev)
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        got = pred[k][:, :, pos].cpu().numpy()
        err = np.max(np.abs(got - g["out/" + k]))
        assert err < TOL, (k, err)
        s = float(pred[k].double().sum().cpu())
        assert abs(s - float(g["outsum/" + k])) < 1e-4 * pred[k].numel() ** 0.5 + 1e-2, k
