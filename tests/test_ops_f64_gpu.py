"""The operators' DOUBLE dispatch (the reference: AT_DISPATCH_FLOATING_TYPES in every .cu of pn2_ext) against
the C oracle compiled with scalar_t = double (oracle/Makefile: the same source, -DS4G_ORACLE_F64)."""
import numpy as np
import pytest
import torch

from s4g_release_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _double_oracle(oracle):
    """This module checks the operators' scalar_t = double case: the oracle's double build is opt-in."""
    with oracle.double_dispatch():
        yield


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def F():
    from s4g_release_amd import functions
    functions.set_distance_mode("strict")
    return functions


def _clouds(kind, B, N, seed=0):
    if kind == "lattice":                       # exact ties: the tie rule decides
        rng = np.random.default_rng(seed)
        return rng.integers(0, 6, size=(B, 3, N)).astype(np.float64) * 0.125
    # double coordinates that are NOT representable in float: the double path must not round them
    base = synth.make_batch(list(range(seed, seed + B)), N).astype(np.float64)
    return base + np.random.default_rng(seed).standard_normal(base.shape) * 1e-9


@pytest.mark.parametrize("kind,N,M", [("tabletop", 3000, 700), ("lattice", 1500, 400), ("tabletop", 16, 16),
                                      ("lattice", 64, 64), ("tabletop", 700, 1)])
def test_fps_f64(F, oracle, dev, kind, N, M):
    pts = _clouds(kind, 2, N)
    got = F.farthest_point_sample(_t(pts, dev), M)
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), oracle.fps(pts, M))
    if N <= 1500:
        assert np.array_equal(got.cpu().numpy(), oracle.fps_literal(pts, M))      # the literal 512-thread emulation agrees


@pytest.mark.parametrize("kind,N,M,r,K", [("tabletop", 4000, 500, 0.03, 32), ("lattice", 2000, 300, 0.25, 64),
                                          ("tabletop", 4000, 200, 0.2, 16), ("tabletop", 100, 10, 0.05, 200)])
def test_ball_query_f64(F, oracle, dev, kind, N, M, r, K):
    pts = _clouds(kind, 2, N, seed=3)
    ctr = oracle.gather_points(pts, oracle.fps(pts, M))
    ctr[0, :, 0] += 50.0                                         # an empty ball: count 0, indices 0
    idx, cnt = F.ball_query(_t(pts, dev), _t(ctr, dev), r, K)
    ridx, rcnt = oracle.ball_query(pts, ctr, r, K)
    assert np.array_equal(cnt.cpu().numpy(), rcnt) and np.array_equal(idx.cpu().numpy(), ridx)
    assert rcnt[0, 0] == 0 and (0 < rcnt).any() and (rcnt < K).any()
    i2, c2, g2 = F.query_and_group(_t(pts, dev), _t(ctr, dev), r, K)
    assert np.array_equal(i2.cpu().numpy(), ridx) and g2.dtype == torch.float64
    assert np.array_equal(g2.cpu().numpy(), oracle.group_points(pts, ridx))


@pytest.mark.parametrize("kind,N1,N2", [("tabletop", 3000, 500), ("lattice", 800, 90), ("tabletop", 5, 3)])
def test_three_nn_and_interpolate_f64(F, oracle, dev, kind, N1, N2):
    q = _clouds(kind, 2, N1, seed=5)
    k = oracle.gather_points(q, oracle.fps(q, N2)) if kind == "tabletop" else _clouds(kind, 2, N2, seed=6)
    idx, d2 = F.search_nn_distance(_t(q, dev), _t(k, dev), 3)
    ridx, rd2 = oracle.three_nn(q, k)
    assert d2.dtype == torch.float64
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)
    w = F.interp_weights(d2)
    rw = oracle.interp_weights(rd2)
    assert np.allclose(w.cpu().numpy(), rw, rtol=1e-15, atol=0)
    feat = np.random.default_rng(1).standard_normal((2, 7, N2))
    out = F.feature_interpolate(_t(feat, dev), idx, _t(rw, dev))
    assert out.dtype == torch.float64
    assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(feat, ridx, rw))


def test_group_gather_and_backward_f64(F, oracle, dev):
    rng = np.random.default_rng(2)
    feat = rng.standard_normal((2, 5, 300))
    idx = rng.integers(0, 300, size=(2, 40, 8))
    x = _t(feat, dev).requires_grad_(True)
    out = F.group_points(x, _t(idx, dev))
    assert out.dtype == torch.float64 and np.array_equal(out.detach().cpu().numpy(), oracle.group_points(feat, idx))
    g = rng.standard_normal(out.shape)
    out.backward(_t(g, dev))
    assert np.allclose(x.grad.cpu().numpy(), oracle.group_points_backward(g, idx, 300), rtol=1e-13, atol=1e-13)
    gi = rng.integers(0, 300, size=(2, 33))
    assert np.array_equal(F.gather_points(_t(feat, dev), _t(gi, dev)).cpu().numpy(), oracle.gather_points(feat, gi))
    # interpolation backward (atomic scatter: order of the sums is not fixed)
    nidx = rng.integers(0, 300, size=(2, 50, 3))
    w = rng.random((2, 50, 3))
    y = _t(feat, dev).requires_grad_(True)
    o = F.feature_interpolate(y, _t(nidx, dev), _t(w, dev))
    go = rng.standard_normal(o.shape)
    o.backward(_t(go, dev))
    assert np.allclose(y.grad.cpu().numpy(), oracle.three_interpolate_backward(go, nidx, w, 300), rtol=1e-13, atol=1e-13)


def test_fmad_contract_f64(F, oracle, dev):
    pts = _clouds("tabletop", 2, 2500, seed=9)
    try:
        F.set_distance_mode("fmad")
        idx = F.farthest_point_sample(_t(pts, dev), 300).cpu().numpy()
        assert np.array_equal(idx, oracle.fps(pts, 300, fmad=1))
        ctr = oracle.gather_points(pts, idx)
        bi, bc = F.ball_query(_t(pts, dev), _t(ctr, dev), 0.04, 24)
        ri, rc = oracle.ball_query(pts, ctr, 0.04, 24, fmad=1)
        assert np.array_equal(bi.cpu().numpy(), ri) and np.array_equal(bc.cpu().numpy(), rc)
        ni, nd = F.search_nn_distance(_t(pts, dev), _t(ctr, dev), 3)
        rni, rnd = oracle.three_nn(pts, ctr, fmad=1)
        assert np.array_equal(ni.cpu().numpy(), rni) and np.array_equal(nd.cpu().numpy(), rnd)
    finally:
        F.set_distance_mode("strict")


def test_mixed_dtypes_are_refused_and_float_is_untouched(F, dev):
    pts = _t(_clouds("tabletop", 1, 500), dev)
    with pytest.raises(RuntimeError):
        F.ball_query(pts, pts.float(), 0.05, 8)
    with pytest.raises(RuntimeError):
        F.search_nn_distance(pts.float(), pts, 3)
    assert F.farthest_point_sample(pts.float(), 50).dtype == torch.int64


def test_reference_shaped_model_in_double(dev):
    """`net.double()` through the reference-shaped modules: the operators accept it (the reference's would),
    and the result is the float model's up to float round-off."""
    from s4g_release_amd.model import PointNet2, randomize_bn_
    cfg = dict(score_classes=3, num_centroids=(256, 64, 16), radius=(0.05, 0.12, 0.4), num_neighbours=(16, 16, 16),
               sa_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128)), fp_channels=((128, 128), (64, 64), (32, 32, 32)),
               num_fp_neighbours=(3, 3, 3), seg_channels=(64, 32, 32, 16), num_removal_directions=5, dropout_prob=0.5)
    from tests import golden_util as GU
    torch.manual_seed(5)
    pts = torch.from_numpy(synth.make_batch([1, 2], 1024)).to(dev)
    net = GU.calibrated(PointNet2(**cfg).to(dev), 6, pts)
    with torch.no_grad():
        a = net({"scene_points": pts})
        b = net.double()({"scene_points": pts.double()})
    for k in a:
        assert b[k].dtype == torch.float64
        # calibrated weights (outputs up to +-4 that depend on the input): fp32 through the modules vs the same in double
        assert (a[k].double() - b[k]).abs().max().item() < GU.WIRING_TOL64 * max(1.0, b[k].abs().max().item()), k
