"""GPU parity against the CALIBRATED golden fixtures (tools/gen_golden_calib.py: the reference's own PointNet2,
BatchNorm statistics calibrated through the reference's own modules).  Unlike the first fixture family, whose
outputs are per-channel constants, these depend on the input at every point: the four outputs AND every level's
feature tensor are compared, so a wrong centroid assignment, channel order or concat order anywhere in the HIP path
shows (the CPU suite's sabotage tests prove the fixtures see such errors at >= 100 x the tolerance).

Tolerance: 1e-4 x max(1, max|ref|) per tensor (BASELINE.json north_star's "1e-4 fp32", relative to the tensor's
scale: the outputs reach +-5).  The fixture's own distance from float64 arithmetic -- torch's CPU fp32 kernels under
the reference network -- is 2.4e-5 .. 4.9e-5 of scale (stored in the fixture as margin/*, printed by the
generator); it is not gamma's spread that sets it (2.6e-5 .. 3.3e-5 with gamma = 1) but the re-normalisation of
every layer, which is what a trained network does.  Index tensors: bit-exact."""
import numpy as np
import pytest
import torch

from tests import golden_util as GU

pytestmark = pytest.mark.gpu
PRECISIONS = ["f16x2", "bf16x3", "fp32"]


def _hook_levels(net, store):
    for i, m in enumerate(net.sa_modules):
        m.register_forward_hook(lambda mod, a, out, i=i: store.__setitem__("sa%d" % i, out[1]))
    for i, m in enumerate(net.fp_modules):
        m.register_forward_hook(lambda mod, a, out, i=i: store.__setitem__("fp%d" % i, out))


def _np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def _rel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - ref).max()) / max(1.0, float(np.abs(ref).max()))


def _small_net():
    from s4g_release_amd.model import PointNet2
    g = GU.load("pn2_calib_small.npz")
    net = PointNet2(**GU.small_config(g))
    net.load_state_dict(GU.small_state_dict(g), strict=True)
    return g, net.eval()


def _check_small(g, pred, feats, inter=None):
    if inter is not None:
        for li in range(3):
            for n in ("fps", "ball", "cnt", "nn"):
                assert np.array_equal(inter["%s%d" % (n, li)].cpu().numpy().astype(np.int64),
                                      g["%s%d" % (n, li)].astype(np.int64)), (n, li)
    worst, exact = {}, {}
    for k in GU.HEADS:
        worst[k] = _rel(pred[k], g["out/" + k])
        exact[k] = _rel(pred[k], g["out64/" + k])
    print("vs float64:", {k: "%.1e" % v for k, v in exact.items()})
    for lv, a in feats.items():
        assert a.shape == g["feat/" + lv].shape, (lv, a.shape)
        worst[lv] = _rel(a, g["feat/" + lv])
    print({k: "%.1e" % v for k, v in worst.items()})
    # Reduced config (16 .. 128 channels, calibrated over 4 096 points): the fixture -- torch's CPU fp32 kernels under
    # the reference network -- is itself 3.5e-5 .. 6.0e-5 of scale from float64 arithmetic (margin/*, stored by the
    # generator).  So: every path within 1e-4 of scale of the EXACT result, and within 1e-4 + the fixture's own
    # distance of the fixture (triangle).  Measured vs float64 / vs fixture, worst head: modules 6.9e-5 / 7.2e-5,
    # f16x2 5.5e-5 / 1.03e-4, bf16x3 3.9e-5 / 6.3e-5, fp32 4.5e-5 / 8.3e-5.  Level features: strict.
    for k, v in exact.items():
        assert v < GU.CALIB_TOL, (k, "vs float64", v)
    for k, v in worst.items():
        slack = float(g["margin/" + k][0]) if k in GU.HEADS else 0.0
        assert v < GU.CALIB_TOL + slack, (k, v)
    return worst


def test_reference_shaped_modules_small_calibrated(dev):
    """The drop-in level (reference-shaped modules on the HIP operators): outputs + all six level features."""
    g, net = _small_net()
    net = net.to(dev)
    feats = {}
    _hook_levels(net, feats)
    with torch.no_grad():
        pred = net({"scene_points": torch.from_numpy(g["points"]).to(dev)})
    _check_small(g, _np(pred), _np(feats))
    assert sorted(feats) == sorted(GU.LEVELS)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fused_small_calibrated(dev, precision):
    from s4g_release_amd.fused import FusedPointNet2
    g, net = _small_net()
    run = FusedPointNet2(net.to(dev), precision=precision)
    pred, inter = run({"scene_points": torch.from_numpy(g["points"]).to(dev)}, return_intermediates=True)
    feats = {k[5:]: v for k, v in inter.items() if k.startswith("feat_")}
    assert {"sa0", "sa1", "sa2"} <= set(feats), sorted(feats)
    _check_small(g, _np(pred), _np(feats), inter)


@pytest.fixture(scope="module")
def full():
    g = GU.load("pn2_calib_full.npz")
    return g, GU.calib_full_model(g), GU.calib_scenes(g)


def _check_indices_full(g, scene, inter):
    for li in range(3):
        for n in ("fps", "ball", "cnt", "nn"):
            got = inter["%s%d" % (n, li)].cpu().numpy().astype(np.int64)
            assert GU.sha(got) == str(g["%s%d_sha256/%s" % (n, li, scene)]), (scene, n, li)


@pytest.mark.parametrize("scene", ["tabletop", "real"])
def test_reference_shaped_modules_full_calibrated(dev, full, scene):
    g, net, scenes = full
    net = net.to(dev)
    feats = {}
    _hook_levels(net, feats)
    with torch.no_grad():
        pred = net({"scene_points": torch.from_numpy(scenes[scene]).to(dev)})
    worst = GU.calib_compare_full(g, scene, _np(pred), _np(feats))
    print(scene, "modules", {k: "%.1e" % v for k, v in worst.items()})


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("scene", ["tabletop", "real"])
def test_fused_full_calibrated(dev, full, scene, precision):
    """Shipped config at full size: outputs at 256 positions + float64 sums over all 25 600 points, every level
    feature tensor the fused path materialises (all three SA levels and FP1 on the shipped f16x2 path -- FP2 / FP3
    live only inside their consumers' launches; the fp32 path writes all six), indices by sha256."""
    from s4g_release_amd.fused import FusedPointNet2
    g, net, scenes = full
    run = FusedPointNet2(net.to(dev), precision=precision)
    pred, inter = run({"scene_points": torch.from_numpy(scenes[scene]).to(dev)}, return_intermediates=True)
    _check_indices_full(g, scene, inter)
    feats = {k[5:]: v for k, v in inter.items() if k.startswith("feat_")}
    need = GU.LEVELS if precision == "fp32" else ("sa0", "sa1", "sa2", "fp0")
    worst = GU.calib_compare_full(g, scene, _np(pred), _np(feats), need_levels=need)
    print(scene, precision, {k: "%.1e" % v for k, v in worst.items()})


def test_fused_full_calibrated_batched_and_graph(dev, full):
    """Both scenes as ONE batch (the fixture's own batch), eager and as a replayed HIP graph."""
    from s4g_release_amd.fused import FusedPointNet2
    g, net, scenes = full
    run = FusedPointNet2(net.to(dev))
    names = [str(n) for n in g["scenes"]]
    pts = torch.from_numpy(np.concatenate([scenes[n] for n in names], axis=0)).to(dev)
    pred = _np(run({"scene_points": pts}))
    graphed = run.graph({"scene_points": pts})
    pred_g = _np(graphed({"scene_points": pts}))
    for s, n in enumerate(names):
        for p in (pred, pred_g):
            GU.calib_compare_full(g, n, {k: v[s:s + 1] for k, v in p.items()}, {}, need_levels=())


def test_sabotaged_device_path_is_caught(dev, full):
    """The device-side twin of the CPU sabotage tests: the reference-shaped modules with SA1's features rolled by
    one centroid must miss the fixture by >= 100 x the tolerance (it passed the old fixtures at 4.8e-7)."""
    g, net, scenes = full
    net = net.to(dev)
    h = net.sa_modules[0].register_forward_hook(lambda mod, a, out: (out[0], torch.roll(out[1], 1, dims=2)))
    try:
        with torch.no_grad():
            pred = _np(net({"scene_points": torch.from_numpy(scenes["real"]).to(dev)}))
    finally:
        h.remove()
    pos = g["positions"]
    worst = max(_rel(pred[k][0][:, pos], g["out/real/" + k]) for k in GU.HEADS)
    assert worst >= 100 * GU.CALIB_TOL, worst
