"""CPU tests against the golden fixtures captured from the reference's Python
network (tools/gen_golden.py): the oracle forward reproduces them, and the
product model exposes the reference's state_dict layout and regenerates the
golden weights from the seed."""
import numpy as np
import pytest
import torch

from tests import golden_util as GU


def test_oracle_forward_reproduces_reference_small():
    from oracle import pn2_forward
    g = GU.load("pn2_small.npz")
    cfg = GU.small_config(g)
    out, inter = pn2_forward.forward(GU.small_state_dict(g), g["points"], cfg["num_centroids"],
                                     cfg["radius"], cfg["num_neighbours"],
                                     return_intermediates=True)
    for li in range(3):
        assert np.array_equal(inter["fps%d" % li], g["fps%d" % li])
        assert np.array_equal(inter["ball%d" % li], g["ball%d" % li])
        assert np.array_equal(inter["cnt%d" % li], g["cnt%d" % li])
        assert np.array_equal(inter["nn%d" % li], g["nn%d" % li])
        assert np.array_equal(inter["nnd%d" % li], g["nnd%d" % li])
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        assert out[k].shape == g["out/" + k].shape
        assert np.max(np.abs(out[k] - g["out/" + k])) < 1e-5, k


def test_product_model_state_dict_layout_small():
    from s4g_release_amd.model import PointNet2, randomize_bn_
    g = GU.load("pn2_small.npz")
    cfg = GU.small_config(g)
    torch.manual_seed(int(g["seed"]))
    net = PointNet2(**cfg)
    randomize_bn_(net, int(g["seed"]) + 1)
    ref_sd = GU.small_state_dict(g)
    sd = net.state_dict()
    assert sorted(sd.keys()) == sorted(ref_sd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), k
    # same construction order => same RNG stream => identical weights
    assert GU.state_dict_sha256(sd) == str(g["state_dict_sha256"])
    net.load_state_dict(ref_sd, strict=True)


def test_product_model_matches_shipped_config_layout():
    from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, strip_module_prefix
    g = GU.load("pn2_full.npz")
    net = build_pointnet2_cls(S4GConfig())
    sd = net.state_dict()
    assert len(sd) == 200
    assert sorted(sd.keys()) == [str(k) for k in g["state_dict_keys"]]
    shapes = {str(k): str(s) for k, s in zip(g["state_dict_keys"], g["state_dict_shapes"])}
    for k, v in sd.items():
        assert repr(tuple(v.shape)) == shapes[k], k
    assert sum(p.numel() for p in net.parameters()) == int(g["num_params"]) == 6632213
    wrapped = {"module." + k: v for k, v in sd.items()}
    assert sorted(strip_module_prefix(wrapped)) == sorted(sd)


def test_full_config_weights_regenerate_from_seed():
    g = GU.load("pn2_full.npz")
    net = GU.build_full_model(int(g["seed"]))
    assert GU.state_dict_sha256(net.state_dict()) == str(g["state_dict_sha256"])


def test_full_config_input_regenerates():
    from s4g_release_amd import synth
    g = GU.load("pn2_full.npz")
    assert GU.sha(synth.make_batch([int(g["scene_id"])], 25600)) == str(g["points_sha256"])


@pytest.mark.parametrize("fixture", ["pn2_real.npz", "pn2_real_replace.npz"])
def test_real_scene_fixture_geometry_with_oracle(fixture):
    """Oracle operators on the reference's sample scene (fixtures pn2_real.npz: a seeded subsample;
    pn2_real_replace.npz: drawn WITH replacement like the harness does, 19 986 distinct points among the
    25 600): the SA1 sampling and grouping indices the reference's modules requested."""
    from oracle import oracle as O
    g = GU.load(fixture)
    pts = g["points"]
    assert pts.shape == (1, 3, 25600) and int(g["source_points"]) == 48902
    if "replace" in fixture:
        assert len(np.unique(pts[0].T, axis=0)) <= int(g["distinct_points"]) == 19986
    fps = O.fps(pts, 5120)
    assert np.array_equal(fps[:, :256], g["fps0_head"]) and GU.sha(fps) == str(g["fps0_sha256"])
    ctr = O.gather_points(pts, fps)
    idx, cnt = O.ball_query(pts, ctr, 0.02, 64)
    assert GU.sha(idx) == str(g["ball0_sha256"]) and GU.sha(cnt) == str(g["cnt0_sha256"])
