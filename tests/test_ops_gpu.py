"""GPU parity tests: every operator of libs4g_hip.so, called through the
Python operator API (ctypes -> C ABI -> HIP kernels), against the CPU oracle on
the same seeded inputs.  Integer results must be bit-exact; fp32 results of the
canonical-arithmetic ops (squared distances, weights, interpolation) too."""
import numpy as np
import pytest
import torch

from s4g_release_amd import synth
from tests.conftest import with_variants

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _quantized(rng, B, N, levels=4, scale=0.25):
    return rng.integers(0, levels, size=(B, 3, N)).astype(np.float32) * np.float32(scale)


@pytest.fixture(scope="module")
def F():
    from s4g_release_amd import functions
    functions.set_distance_mode("strict")
    return functions


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("N,M", [(3, 3), (15, 7), (16, 16), (64, 64), (100, 37), (513, 64),
                                 (1024, 256), (1500, 41), (5120, 1024), (6000, 100),
                                 (12000, 64), (20000, 50)])
def test_fps_small_and_ties(F, oracle, dev, N, M):
    rng = np.random.default_rng(N + M)
    for pts in (_quantized(rng, 2, N), rng.random((2, 3, N), dtype=np.float32)):
        got = F.farthest_point_sample(_t(pts, dev), M).cpu().numpy()
        assert got.dtype == np.int64
        assert np.array_equal(got, oracle.fps(pts, M))


@pytest.mark.parametrize("variant", ["tabletop-v1", "dup-heavy", "uniform-box"])
def test_fps_full_size_sa1(F, oracle, dev, variant):
    pts = synth.make_batch([0, 1], 25600, variant=variant)
    got = F.farthest_point_sample(_t(pts, dev), 5120).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 5120))


@pytest.mark.parametrize("N,M,variant", [(25600, 5120, "tabletop-v1"), (25600, 700, "dup-heavy"),
                                         (20000, 300, "uniform-box"), (6000, 1500, "tabletop-v1"),
                                         (700, 700, "dup-heavy")])
def test_fps_pruned_variant_is_exact(F, oracle, dev, monkeypatch, N, M, variant):
    """Group-pruned kernel (Morton order + per-group bounding boxes; default above 10 240
    points, forced here for the small sizes too): skipping a group is exact by monotonicity
    of the rounded distance, ties included.  The full scan must agree as well."""
    monkeypatch.setenv("S4G_FPS_MODE", "pruned")
    pts = synth.make_batch([3, 5], N, variant=variant)
    got = F.farthest_point_sample(_t(pts, dev), M).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, M))
    monkeypatch.setenv("S4G_FPS_MODE", "dense")
    assert np.array_equal(F.farthest_point_sample(_t(pts, dev), M).cpu().numpy(), got)


@pytest.mark.parametrize("mode", with_variants(["auto"], ["cluster", "hybrid"]))
def test_fps_hybrid_kernel_large_cloud(F, oracle, dev, monkeypatch, mode):
    """25 600 < N <= 51 200.  auto (default): one workgroup per scene, pruned, min-distances in
    registers and the coordinates of touched groups read from the Morton-sorted records in L2;
    cluster: two workgroups per scene, each with half of the points in registers, the local
    winners exchanged through L2 every step; hybrid: one workgroup, full scan, x + min-distance
    in registers, y / z re-read from L2."""
    monkeypatch.setenv("S4G_FPS_MODE", mode)
    pts = synth.make_batch([2], 51200)
    got = F.farthest_point_sample(_t(pts, dev), 300).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 300))
    for n, variant in ((30000, "dup-heavy"), (48902, "tabletop-v1"), (25601, "uniform-box")):
        pts = synth.make_batch([6], n, variant=variant)
        got = F.farthest_point_sample(_t(pts, dev), 200).cpu().numpy()
        assert np.array_equal(got, oracle.fps(pts, 200)), (n, variant)


@pytest.mark.parametrize("mode", with_variants(["auto"], ["cluster"]))
@pytest.mark.parametrize("variant", ["tabletop-v1", "dup-heavy"])
def test_fps_cluster_full_size_batch_ties_and_fmad(F, oracle, dev, monkeypatch, variant, mode):
    """configs[4] geometry: FPS 51 200 -> 5 120 with ALL 5 119 steps, three scenes, exact ties
    (duplicate-heavy cloud; in cluster mode they straddle the two cooperating workgroups), both
    arithmetic contracts; the int32 + centroid-gather entry point of the fast path agrees."""
    monkeypatch.setenv("S4G_FPS_MODE", mode)
    pts = synth.make_batch([0, 1, 7], 51200, variant=variant)
    got = F.farthest_point_sample(_t(pts, dev), 5120).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 5120))
    q = _quantized(np.random.default_rng(1), 2, 40000, levels=24, scale=0.03125)
    assert np.array_equal(F.farthest_point_sample(_t(q, dev), 400).cpu().numpy(), oracle.fps(q, 400))
    try:
        F.set_distance_mode("fmad")
        got = F.farthest_point_sample(_t(pts[:1], dev), 600).cpu().numpy()
    finally:
        F.set_distance_mode("strict")
    assert np.array_equal(got, oracle.fps(pts[:1], 600, fmad=1))


def test_fps_clouds_subsampled_with_replacement_all_picks(F, oracle, dev):
    """The reference's own harness draws its 25 600 points WITH replacement (grasp_proposal_test.py:26-29), so exact
    copies are the normal input.  Copies of the winner inside its group fall to zero with it: they no longer veto the
    exchange's further picks (round 4: FPS of such clouds 7.5 -> 5.6 ms per 16 scenes), and every one of the 5 119
    picks still is the oracle's, including which copy's index is reported."""
    rng = np.random.default_rng(29)
    src = synth.make_batch([5], 48902)[0]
    pts = np.stack([src[:, rng.integers(0, 48902, size=25600)] for _ in range(2)]).astype(np.float32)
    assert len(np.unique(pts[0].T, axis=0)) < 0.85 * 25600          # ~21 % of the draws repeat an earlier one
    got = F.farthest_point_sample(_t(pts, dev), 5120).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 5120))
    dup = synth.make_batch([1], 25600, variant="dup-heavy")
    got = F.farthest_point_sample(_t(dup, dev), 5120).cpu().numpy()
    assert np.array_equal(got, oracle.fps(dup, 5120))
    big = np.stack([src[:, rng.integers(0, 48902, size=51200)]]).astype(np.float32)   # the L2-resident kernel
    got = F.farthest_point_sample(_t(big, dev), 3000).cpu().numpy()
    assert np.array_equal(got, oracle.fps(big, 3000))


def test_fps_streaming_fallback_very_large_cloud(F, oracle, dev):
    """N > 65 535: the streaming kernel (min-distances in the workspace); M < 64 takes it at any size."""
    pts = synth.make_batch([2], 70000)
    got = F.farthest_point_sample(_t(pts, dev), 100).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 100))
    pts = synth.make_batch([2], 60000)
    got = F.farthest_point_sample(_t(pts, dev), 40).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 40))


@pytest.mark.parametrize("N,M,variant", [(51201, 900, "tabletop-v1"), (60000, 1500, "dup-heavy"),
                                         (65535, 700, "lattice"), (65535, 2000, "tabletop-v1")])
def test_fps_l2_pruned_kernel_128_slots(F, oracle, dev, N, M, variant):
    """51 200 < N <= 65 535 (round 4): the L2-resident pruned kernel with 128 min-distances per lane instead of
    the streaming fallback (16 scenes of 65 535 points -> 5 120: 6.6 ms instead of ~75); both distance contracts."""
    pts = synth.make_batch([3, 9], N, variant=variant)
    got = F.farthest_point_sample(_t(pts, dev), M).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, M))
    try:
        F.set_distance_mode("fmad")
        got = F.farthest_point_sample(_t(pts[:1], dev), 300).cpu().numpy()
    finally:
        F.set_distance_mode("strict")
    assert np.array_equal(got, oracle.fps(pts[:1], 300, fmad=1))


def test_fps_fmad_mode_pruned_size(F, oracle, dev):
    pts = synth.make_batch([4], 12000)
    try:
        F.set_distance_mode("fmad")
        got = F.farthest_point_sample(_t(pts, dev), 700).cpu().numpy()
    finally:
        F.set_distance_mode("strict")
    assert np.array_equal(got, oracle.fps(pts, 700, fmad=1))


def test_fps_fmad_mode(F, oracle, dev):
    pts = synth.make_batch([4], 4096)
    try:
        F.set_distance_mode("fmad")
        got = F.farthest_point_sample(_t(pts, dev), 512).cpu().numpy()
    finally:
        F.set_distance_mode("strict")
    assert np.array_equal(got, oracle.fps(pts, 512, fmad=1))


def test_fps_argument_errors(F, dev):
    pts = torch.zeros(1, 3, 4, device=dev)
    with pytest.raises(RuntimeError):
        F.farthest_point_sample(pts, 5)
    with pytest.raises(RuntimeError):
        F.farthest_point_sample(pts, 0)
    with pytest.raises(RuntimeError):
        F.farthest_point_sample(torch.zeros(1, 4, 4, device=dev), 2)


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize("variant", ["tabletop-v1", "dup-heavy", "uniform-box"])
@pytest.mark.parametrize("N,M,r,K", [(25600, 5120, 0.02, 64), (5120, 1024, 0.08, 64),
                                     (1024, 256, 0.32, 64), (777, 33, 0.05, 7),
                                     # radii far above the point spacing: balls whose 27 cells hold a large share of
                                     # the cloud take the index-order scan inside the grid kernel (round 4)
                                     (25600, 2048, 0.2, 64), (25600, 1024, 0.06, 128), (12000, 700, 1.5, 16)])
def test_ball_query_matches_oracle(F, oracle, dev, variant, N, M, r, K):
    pts = synth.make_batch([0, 5], N, variant=variant)
    ctr = oracle.gather_points(pts, oracle.fps(pts, M))
    idx, cnt = F.ball_query(_t(pts, dev), _t(ctr, dev), r, K)
    ridx, rcnt = oracle.ball_query(pts, ctr, r, K)
    assert idx.dtype == torch.int64 and cnt.dtype == torch.int64
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(idx.cpu().numpy(), ridx)


@pytest.fixture
def bq_mode(monkeypatch):
    def set_mode(mode):
        monkeypatch.setenv("S4G_BQ_MODE", mode)
    return set_mode


@pytest.mark.parametrize("mode", with_variants(["grid", "scan"], ["cell"]))
@pytest.mark.parametrize("variant,N,M,r,K", [
    ("tabletop-v1", 25600, 5120, 0.02, 64),    # SA1
    ("tabletop-v1", 25600, 2000, 0.01, 32),    # scene wider than 32 cells: toroidal aliasing
    ("dup-heavy", 25600, 5120, 0.02, 64),      # exact ties, > K hits
    ("uniform-box", 25600, 5120, 0.02, 64),    # ~1 hit per ball, all padded
    ("tabletop-v1", 3000, 777, 0.05, 16),      # small cloud forced through the grid
    ("tabletop-v1", 51200, 1024, 0.02, 100),   # K > 64, 51 200 points
    ("tabletop-v1", 4096, 512, 0.6, 64),       # ball as big as the scene
])
def test_ball_query_grid_and_scan_paths(F, oracle, dev, bq_mode, mode, variant, N, M, r, K):
    bq_mode(mode)
    pts = synth.make_batch([2, 9], N, variant=variant)
    ctr = oracle.gather_points(pts, oracle.fps(pts, M))
    ctr[0, :, 0] += 5.0                      # one centroid far from every point: empty ball
    idx, cnt = F.ball_query(_t(pts, dev), _t(ctr, dev), r, K)
    ridx, rcnt = oracle.ball_query(pts, ctr, r, K)
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert cnt[0, 0].item() == 0 and (idx[0, 0] == 0).all()


@pytest.mark.parametrize("mode", with_variants(["grid"], ["cell"]))
def test_ball_query_grid_out_of_range_scene_falls_back(F, oracle, dev, bq_mode, mode):
    """A scene spanning > 4096 cells trips the exactness flag: its centroids take
    the index-order scan inside the grid kernel; the other scene stays on the grid."""
    bq_mode(mode)
    pts = synth.make_batch([0, 1], 9000)
    pts[1, 0, 17] += 500.0                   # 500 m / 0.02 m = 25 000 cells
    pts[1, 2, 4000] = np.float32(-3e30)
    ctr = oracle.gather_points(pts, oracle.fps(pts, 300))
    idx, cnt = F.ball_query(_t(pts, dev), _t(ctr, dev), 0.02, 48)
    ridx, rcnt = oracle.ball_query(pts, ctr, 0.02, 48)
    assert np.array_equal(cnt.cpu().numpy(), rcnt) and np.array_equal(idx.cpu().numpy(), ridx)
    i2, c2, g2 = F.query_and_group(_t(pts, dev), _t(ctr, dev), 0.02, 48)
    assert np.array_equal(i2.cpu().numpy(), ridx)
    assert np.array_equal(g2.cpu().numpy(), oracle.group_points(pts, ridx))


@pytest.mark.parametrize("mode", with_variants(["auto", "scan", "grid"], ["cell"]))
@pytest.mark.parametrize("N,M,r", [(25600, 5120, 0.02), (5120, 1024, 0.08)])
def test_query_and_group_equals_operator_pair(F, oracle, dev, bq_mode, mode, N, M, r):
    bq_mode(mode)
    pts = synth.make_batch([3, 4], N)
    ctr = oracle.gather_points(pts, oracle.fps(pts, M))
    idx, cnt, grouped = F.query_and_group(_t(pts, dev), _t(ctr, dev), r, 64)
    ridx, rcnt = oracle.ball_query(pts, ctr, r, 64)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(grouped.cpu().numpy(), oracle.group_points(pts, ridx))
    assert tuple(grouped.shape) == (2, 3, M, 64) and grouped.is_contiguous()


def test_ball_query_empty_balls_and_padding(F, oracle, dev):
    pts = np.zeros((1, 3, 10), dtype=np.float32)
    pts[0, 0, :] = np.arange(10)
    ctr = np.array([[[4.0, 100.0], [0.0, 0.0], [0.0, 0.0]]], dtype=np.float32)
    idx, cnt = F.ball_query(_t(pts, dev), _t(ctr, dev), 1.5, 6)
    assert cnt.cpu().tolist() == [[3, 0]]
    assert idx.cpu()[0, 0].tolist() == [3, 4, 5, 3, 3, 3]
    assert idx.cpu()[0, 1].tolist() == [0] * 6


# ------------------------------------------------------------------ group / gather
@pytest.mark.parametrize("C,N,M,K", [(3, 25600, 5120, 64), (256, 5120, 1024, 64), (5, 100, 7, 3),
                                     (68, 1000, 131, 33), (16, 300, 64, 64), (132, 77, 257, 16)])
def test_group_points_matches_oracle(F, oracle, dev, C, N, M, K):
    rng = np.random.default_rng(C)
    feat = rng.standard_normal((2, C, N)).astype(np.float32)
    index = rng.integers(0, N, size=(2, M, K))
    out = F.group_points(_t(feat, dev), _t(index, dev))
    assert out.is_contiguous() and tuple(out.shape) == (2, C, M, K)
    assert np.array_equal(out.cpu().numpy(), oracle.group_points(feat, index))
    # the caller mutates the result in place (modules.py:44): must be a fresh tensor
    out -= 1.0


def test_gather_points_matches_oracle(F, oracle, dev):
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((3, 3, 999)).astype(np.float32)
    index = rng.integers(0, 999, size=(3, 77))
    out = F.gather_points(_t(pts, dev), _t(index, dev))
    assert np.array_equal(out.cpu().numpy(), oracle.gather_points(pts, index))


def test_group_points_backward(F, oracle, dev):
    rng = np.random.default_rng(2)
    feat = torch.from_numpy(rng.standard_normal((2, 6, 50)).astype(np.float32)).to(dev)
    feat.requires_grad_(True)
    index = rng.integers(0, 50, size=(2, 9, 4))
    g = rng.standard_normal((2, 6, 9, 4)).astype(np.float32)
    out = F.group_points(feat, _t(index, dev))
    out.backward(_t(g, dev))
    ref = oracle.group_points_backward(g, index, 50)
    assert np.allclose(feat.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ 3-NN + interpolate
def _fps_ex(pts, M, dev, want_dist=True, run=None):
    """s4g_fps_gather_ex_i32 -> (idx (B,M) int64, ctr (B,3,M), dist (B,M) or None) as numpy."""
    from s4g_release_amd import _cabi, functions as F
    B, _, N = pts.shape
    x = _t(pts, dev)
    idx = torch.full((B, M), -7, dtype=torch.int32, device=dev)
    ctr = torch.zeros((B, 3, M), dtype=torch.float32, device=dev)
    dist = torch.zeros((B, M), dtype=torch.float32, device=dev) if want_dist else None
    ws, nbytes = F._workspace(_cabi.S4G_OP_FPS, x.device, B, N, M, 0)
    rc = _cabi.lib().s4g_fps_gather_ex_i32(x.data_ptr(), B, N, M, idx.data_ptr(), ctr.data_ptr(),
                                           None if dist is None else dist.data_ptr(),
                                           None if run is None else run.data_ptr(), F._ptr(ws), nbytes,
                                           F._DIST_FLAGS, torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "fps_gather_ex")
    torch.cuda.synchronize()
    return (idx.cpu().numpy().astype(np.int64), ctr.cpu().numpy(),
            None if dist is None else dist.cpu().numpy())


def _prefix_check(ctr, dist, M2, dev):
    from s4g_release_amd import _cabi, functions as F
    B, _, M1 = ctr.shape
    c, d = _t(ctr, dev), _t(dist, dev)
    run = torch.full((B,), 5, dtype=torch.int32, device=dev)
    rc = _cabi.lib().s4g_fps_prefix_check_f32(c.data_ptr(), d.data_ptr(), B, M1, M2, run.data_ptr(),
                                              F._DIST_FLAGS, torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "fps_prefix_check")
    torch.cuda.synchronize()
    return run


def _pick_distances(pts_b, idx_b):
    """D_k = min_{i<k} ((dx*dx)+(dy*dy))+(dz*dz) in fp32, every operation rounded (numpy fp32)."""
    c = pts_b[:, idx_b].astype(np.float32)                       # (3, M)
    M = c.shape[1]
    out = np.full(M, np.inf, np.float32)
    md = np.full(M, np.inf, np.float32)
    for k in range(1, M):
        d = c - c[:, k - 1:k]
        md = np.minimum(md, (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        out[k] = md[k]
    return out


@pytest.mark.parametrize("variant,N,M1,M2", [("tabletop-v1", 25600, 5120, 1024), ("dup-heavy", 25600, 5120, 1024),
                                             ("uniform-box", 12000, 3000, 700), ("tabletop-v1", 5120, 1024, 256),
                                             ("tabletop-v1", 51200, 5120, 1024)])
def test_fps_of_an_fps_ordered_set_is_its_prefix(oracle, dev, variant, N, M1, M2):
    """The next set-abstraction level samples the previous level's centroids (modules.py:80-83), which
    are in pick order: FPS then re-picks their prefix unless two of them tie.  (1) the pick distances
    s4g_fps_gather_ex_i32 reports are the fp32 min-distances of the picks; (2) on these clouds
    s4g_fps_prefix_check_f32 proves the prefix property for every scene; (3) the ORACLE's FPS over the
    centroids is the identity prefix indeed; (4) the conditional call returns exactly that without
    sampling."""
    pts = synth.make_batch([2, 7], N, variant=variant)
    idx, ctr, dist = _fps_ex(pts, M1, dev)
    assert np.array_equal(idx, oracle.fps(pts, M1))
    for b in range(2):
        assert np.array_equal(dist[b], _pick_distances(pts[b], idx[b])), b
    run = _prefix_check(ctr, dist, M2, dev)
    assert run.cpu().tolist() == [0, 0]
    want = oracle.fps(ctr, M2)
    assert np.array_equal(want, np.tile(np.arange(M2), (2, 1)))
    idx2, ctr2, _ = _fps_ex(ctr, M2, dev, want_dist=False, run=run)
    assert np.array_equal(idx2, want) and np.array_equal(ctr2, ctr[:, :, :M2])


def test_fps_prefix_check_fmad_contract_and_unsupported_sizes(F, oracle, dev):
    """The check uses the sampler's arithmetic, also under the fmad contract (S4G_FLAG_FMAD); sizes whose
    kernels cannot report pick distances answer S4G_EUNSUPPORTED before launching anything."""
    from s4g_release_amd import _cabi
    pts = synth.make_batch([4, 6], 12000)
    F.set_distance_mode("fmad")
    try:
        idx, ctr, dist = _fps_ex(pts, 2400, dev)
        assert np.array_equal(idx, oracle.fps(pts, 2400, fmad=1))
        run = _prefix_check(ctr, dist, 600, dev)
        assert run.cpu().tolist() == [0, 0]
        assert np.array_equal(oracle.fps(ctr, 600, fmad=1), np.tile(np.arange(600), (2, 1)))
    finally:
        F.set_distance_mode("strict")
    big = _t(synth.make_batch([1], 70000), dev)
    idx = torch.full((1, 100), -7, dtype=torch.int32, device=dev)
    ctr = torch.zeros((1, 3, 100), dtype=torch.float32, device=dev)
    dist = torch.zeros((1, 100), dtype=torch.float32, device=dev)
    ws, nbytes = F._workspace(_cabi.S4G_OP_FPS, big.device, 1, 70000, 100, 0)
    rc = _cabi.lib().s4g_fps_gather_ex_i32(big.data_ptr(), 1, 70000, 100, idx.data_ptr(), ctr.data_ptr(),
                                           dist.data_ptr(), None, F._ptr(ws), nbytes, F._DIST_FLAGS,
                                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == _cabi.S4G_EUNSUPPORTED and (idx == -7).all()


def test_fps_prefix_check_refuses_ties_and_the_sampler_runs(oracle, dev):
    """Lattice clouds: many points share their distance to the picked set, so the tie rule of
    sampling_kernel.cu:87-105 decides and the next level is NOT the prefix in general.  The check
    must flag such scenes (never claim a prefix the oracle does not produce), and the conditional
    call must then equal the oracle's FPS for flagged and unflagged scenes alike."""
    rng = np.random.default_rng(11)
    flagged = 0
    for trial in range(6):
        pts = _quantized(rng, 3, 4000, levels=12, scale=0.05)
        if trial % 2:
            pts[1] = synth.make_batch([trial], 4000)[0]            # a tie-free scene between two lattices
        M1, M2 = 900, 300
        idx, ctr, dist = _fps_ex(pts, M1, dev)
        assert np.array_equal(idx, oracle.fps(pts, M1))
        run = _prefix_check(ctr, dist, M2, dev)
        want = oracle.fps(ctr, M2)
        for b in range(3):
            if run[b].item() == 0:
                assert np.array_equal(want[b], np.arange(M2)), (trial, b)
        flagged += int((run != 0).sum().item())
        idx2, ctr2, _ = _fps_ex(ctr, M2, dev, want_dist=False, run=run)
        assert np.array_equal(idx2, want), trial
        assert np.array_equal(ctr2, oracle.gather_points(ctr, want)), trial
    assert flagged > 0                                              # the lattice scenes do tie


def _fps_prepass(pts, dev):
    """s4g_fps_prepass_f32 -> (perm (B,N) int64, boxes (B,G,6) fp32) as numpy."""
    from s4g_release_amd import _cabi
    B, _, N = pts.shape
    G = (N + 63) // 64
    x = _t(pts, dev)
    perm = torch.full((B, N), -1, dtype=torch.int32, device=dev)
    box = torch.zeros((B, G, 6), dtype=torch.float32, device=dev)
    rc = _cabi.lib().s4g_fps_prepass_f32(x.data_ptr(), B, N, G, perm.data_ptr(), box.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "fps_prepass")
    torch.cuda.synchronize()
    return perm.cpu().numpy().astype(np.int64), box.cpu().numpy()


def _group_boxes(pts_b, perm_b):
    N = perm_b.shape[0]
    G = (N + 63) // 64
    q = pts_b[:, perm_b]                                   # (3, N) in the pre-pass's order
    out = np.full((G, 6), np.nan, np.float32)
    for g in range(G):
        seg = q[:, 64 * g:64 * g + 64]
        out[g, :3] = seg.min(axis=1)
        out[g, 3:] = seg.max(axis=1)
    return out


@pytest.mark.parametrize("variant,N", [("tabletop-v1", 25600), ("tabletop-v1", 12001), ("uniform-box", 25600),
                                       ("dup-heavy", 20000), ("tabletop-v1", 51200)])
def test_fps_prepass_is_a_permutation_with_exact_tight_boxes(dev, variant, N):
    """The one-launch pre-pass of the pruned FPS kernels (bounding box, cell keys, LDS counting
    sort, group boxes): per scene the output must be a PERMUTATION of 0..N-1 (every later step
    relies on that and on nothing else), every box the exact fp32 min / max of its 64 points, and
    the order spatial: the mean box diagonal at most a fifth of index-order groups' (two fifths on a
    volume-filling cloud)."""
    pts = synth.make_batch([0, 3, 11], N, variant=variant)
    perm, box = _fps_prepass(pts, dev)
    for b in range(pts.shape[0]):
        assert np.array_equal(np.sort(perm[b]), np.arange(N)), b
        want = _group_boxes(pts[b], perm[b])
        assert np.array_equal(box[b], want, equal_nan=True), b
        diag = np.linalg.norm(want[:N // 64, 3:] - want[:N // 64, :3], axis=1).mean()
        base = _group_boxes(pts[b], np.arange(N))
        diag0 = np.linalg.norm(base[:N // 64, 3:] - base[:N // 64, :3], axis=1).mean()
        assert diag < (0.2 if variant != "uniform-box" else 0.4) * diag0, (b, diag, diag0)


def test_fps_prepass_far_outliers_keep_the_spatial_order_and_fps_exact(F, oracle, dev):
    """50 background points 50 m behind a table-top scene (round 4: they stretched the pre-pass's box until the
    scene fell into a handful of cells -- FPS 25 ms instead of 5.4 per 16 scenes).  The box is the range that leaves
    N / 128 points outside per side where that is under half of the extent; outside points share one key.  Still a
    permutation with exact boxes, the regular groups as tight as without the outliers, the outliers in groups of
    their own, and the sample itself bit-exact."""
    N = 25600
    clean = synth.make_batch([2], N)
    pts = clean.copy()
    pts[0, :, 100:150] += 50.0
    pts[0, 2, 150:160] -= 30.0                  # ... and ten below
    perm, box = _fps_prepass(pts, dev)
    perm0, box0 = _fps_prepass(clean, dev)
    assert np.array_equal(np.sort(perm[0]), np.arange(N))
    want = _group_boxes(pts[0], perm[0])
    assert np.array_equal(box[0], want, equal_nan=True)
    diag = np.linalg.norm(want[:N // 64, 3:] - want[:N // 64, :3], axis=1)
    diag0 = np.linalg.norm(box0[0][:N // 64, 3:] - box0[0][:N // 64, :3], axis=1)
    assert np.median(diag) < 1.3 * np.median(diag0), (np.median(diag), np.median(diag0))
    assert (diag > 1.0).sum() <= 3                     # the groups that hold the 60 far points (and a few rim points)
    far = np.zeros(N, bool)
    far[100:160] = True
    groups_with_far = np.unique(np.nonzero(far[perm[0]])[0] // 64)
    assert len(groups_with_far) <= 3
    idx = F.farthest_point_sample(_t(pts, dev), 2048)
    assert np.array_equal(idx.cpu().numpy(), oracle.fps(pts, 2048))


def test_fps_prepass_degenerate_clouds(dev):
    """All points equal; a cloud on a line (two zero extents); a plane; non-finite coordinates:
    still a permutation with exact boxes (the keys only order, they never drop a point)."""
    rng = np.random.default_rng(5)
    N = 3000
    pts = np.zeros((4, 3, N), np.float32)
    pts[0] = 0.25
    pts[1, 0] = rng.random(N, dtype=np.float32)
    pts[2, :2] = rng.random((2, N), dtype=np.float32)
    pts[3] = rng.random((3, N), dtype=np.float32)
    pts[3, 1, 7] = np.inf
    pts[3, 2, 9] = -np.inf
    perm, box = _fps_prepass(pts, dev)
    for b in range(4):
        assert np.array_equal(np.sort(perm[b]), np.arange(N)), b
        assert np.array_equal(box[b], _group_boxes(pts[b], perm[b]), equal_nan=True), b
    with pytest.raises(RuntimeError):
        _fps_prepass(np.zeros((1, 3, 70000), np.float32), dev)     # > 65 535 points


@pytest.mark.parametrize("N1,N2", [(1024, 256), (5120, 1024), (25600, 5120), (70, 3)])
def test_three_nn_matches_oracle(F, oracle, dev, N1, N2):
    pts = synth.make_batch([0, 7], max(N1, 64))[:, :, :N1]
    keys = oracle.gather_points(pts, oracle.fps(pts, N2))
    idx, d2 = F.search_nn_distance(_t(pts, dev), _t(keys, dev), 3)
    ridx, rd2 = oracle.three_nn(pts, keys)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)       # squared, bit-exact
    i2, dd = F.three_nn(_t(pts, dev), _t(keys, dev))
    assert torch.equal(i2, idx) and torch.equal(dd, d2)


@pytest.mark.parametrize("variant,N1,N2", [("dup-heavy", 9000, 4096), ("uniform-box", 12000, 2048),
                                           ("tabletop-v1", 3000, 2500)])
def test_three_nn_operator_api_grid_routing(F, oracle, dev, monkeypatch, variant, N1, N2):
    """search_nn_distance routes N2 >= 2048 through the cell grid with a cell derived from
    the keys' extent: indices AND squared distances must equal the scan's, bit for bit, on
    tie-heavy, volume-filling and surface-like clouds; queries far from every key included."""
    pts = synth.make_batch([4, 9], N1, variant=variant)
    keys = oracle.gather_points(pts, oracle.fps(pts, N2))
    q = pts.copy()
    q[:, :, :7] += np.float32(5.0)                       # isolated queries: fallback scan
    idx, d2 = F.search_nn_distance(_t(q, dev), _t(keys, dev), 3)
    ridx, rd2 = oracle.three_nn(q, keys)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)
    monkeypatch.setenv("S4G_NN_MODE", "scan")
    idx_s, d2_s = F.search_nn_distance(_t(q, dev), _t(keys, dev), 3)
    assert torch.equal(idx_s, idx) and torch.equal(d2_s, d2)


@pytest.mark.parametrize("variant,N1,N2,cell", [
    ("tabletop-v1", 25600, 5120, 0.02),     # FP3 of the shipped config
    ("tabletop-v1", 5120, 2048, 0.08),
    ("dup-heavy", 25600, 5120, 0.02),       # exact distance ties between distinct keys
    ("uniform-box", 25600, 5120, 0.02),     # sparse: most queries need the fallback scan
    ("tabletop-v1", 25600, 5120, 0.004),    # cell far too small: everything falls back
    ("tabletop-v1", 25600, 5120, 0.5),      # cell as big as the scene: all keys are candidates
    ("tabletop-v1", 25600, 5120, -1.0),     # edge chosen on the device from the keys' spacing (the fast path's call)
    ("uniform-box", 25600, 5120, -1.0),     # ... where the SA radius would send nearly every query to the scan
    ("dup-heavy", 5120, 2048, -1.0),
])
def test_three_nn_grid_matches_oracle(F, oracle, dev, variant, N1, N2, cell):
    pts = synth.make_batch([1, 8], N1, variant=variant)
    keys = oracle.gather_points(pts, oracle.fps(pts, N2))
    pts[1, :, 5] = (9.0, 9.0, 9.0)           # an isolated query far from every key
    idx, w = F.three_nn_weights_grid(_t(pts, dev), _t(keys, dev), cell)
    ridx, rd2 = oracle.three_nn(pts, keys)
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)
    assert np.array_equal(w.cpu().numpy(), oracle.interp_weights(rd2))


def test_three_nn_device_chosen_edge_keeps_the_fallback_list_short(F, oracle, dev):
    """The measured edge (1.75 x the mean distance from a key to its third-nearest other key) answers all but a
    handful of queries from the 27 cells on a surface-like AND on a uniformly filled cloud (with the SA radius as the
    edge 98 % of a uniform box's queries took the all-keys scan: 4.8 ms instead of 0.1); the fail count is word 0 of
    the workspace's fail-list header."""
    from s4g_release_amd import _cabi
    for variant in ("tabletop-v1", "uniform-box"):
        pts = synth.make_batch([4], 25600, variant=variant)
        keys = oracle.gather_points(pts, oracle.fps(pts, 5120))
        q, k = _t(pts, dev), _t(keys, dev)
        idx = torch.empty((1, 25600, 3), dtype=torch.int32, device=dev)
        w = torch.empty((1, 25600, 3), dtype=torch.float32, device=dev)
        nbytes = _cabi.lib().s4g_three_nn_grid_workspace_bytes(1, 25600, 5120)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        rc = _cabi.lib().s4g_three_nn_weights_grid_i32(q.data_ptr(), k.data_ptr(), 1, 25600, 5120, 1e-10, -1.0,
                                                       idx.data_ptr(), w.data_ptr(), ws.data_ptr(), nbytes, 0,
                                                       torch.cuda.current_stream().cuda_stream)
        _cabi.check(rc, "three_nn_weights_grid")
        torch.cuda.synchronize()
        grid_bytes = _cabi.lib().s4g_three_nn_grid_header_offset(1, 5120)
        hdr = ws[grid_bytes:grid_bytes + 64].view(torch.int32).cpu()
        ridx, _ = oracle.three_nn(pts, keys)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)
        assert 0 <= int(hdr[0]) < 256, (variant, int(hdr[0]))


def test_three_nn_grid_out_of_range_scene(F, oracle, dev):
    pts = synth.make_batch([2, 3], 6000)
    keys = oracle.gather_points(pts, oracle.fps(pts, 2500))
    keys[0, 0, 100] += 400.0                  # > 4096 cells from the origin: whole scene falls back
    idx, w = F.three_nn_weights_grid(_t(pts, dev), _t(keys, dev), 0.05)
    ridx, rd2 = oracle.three_nn(pts, keys)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)
    assert np.array_equal(w.cpu().numpy(), oracle.interp_weights(rd2))


@pytest.mark.parametrize("cell", [0.01, 0.03])
def test_three_nn_grid_cell_order_seam_and_ragged(F, oracle, dev, cell):
    """The grid search walks the queries in CELL order (they are binned into the keys' grid by the
    build launch): a ragged query count, a scene wider than the 32-cell torus (windows cross the
    seam, far cells alias) and queries whose own cell is outside the exactness range (1e4 away)
    -- every query must be answered exactly once, by the cell walk or by the scan."""
    pts = synth.make_batch([3, 5, 6], 3001)
    keys = oracle.gather_points(pts, oracle.fps(pts, 2100))
    pts[0, :, 17] = (1e4, 0.0, 0.0)
    pts[2, :, 3000] = (-3e3, 2e3, 1e4)
    idx, w = F.three_nn_weights_grid(_t(pts, dev), _t(keys, dev), cell)
    ridx, rd2 = oracle.three_nn(pts, keys)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)
    assert np.array_equal(w.cpu().numpy(), oracle.interp_weights(rd2))


def test_three_nn_ties_and_errors(F, oracle, dev):
    rng = np.random.default_rng(3)
    q, k = _quantized(rng, 2, 300), _quantized(rng, 2, 40)
    idx, d2 = F.search_nn_distance(_t(q, dev), _t(k, dev), 3)
    ridx, rd2 = oracle.three_nn(q, k)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)
    with pytest.raises(RuntimeError):
        F.search_nn_distance(_t(q, dev), _t(k, dev), 2)          # only k == 3
    with pytest.raises(RuntimeError):
        F.search_nn_distance(_t(q, dev), _t(k[:, :, :2], dev), 3)  # N2 >= 3


@pytest.mark.parametrize("N1,N2", [(333, 24), (1000, 100), (1024, 256), (777, 1000), (5120, 1024), (300, 2048)])
@pytest.mark.parametrize("split", ["1", "0"])
def test_three_nn_split_scan_ties_and_sizes(F, oracle, dev, monkeypatch, N1, N2, split):
    """Small key sets take the split scan (S lanes per query, keys in LDS, butterfly merge on
    (distance, key index)); S4G_NN_SPLIT=0 keeps the lane-per-query scan.  Exact ties everywhere
    (coordinates on a coarse lattice, duplicated keys), fmad contract and the int32 + weights
    entry point included."""
    monkeypatch.setenv("S4G_NN_SPLIT", split)
    rng = np.random.default_rng(N1 + N2)
    q, k = _quantized(rng, 2, N1, levels=6), _quantized(rng, 2, N2, levels=6)
    idx, d2 = F.search_nn_distance(_t(q, dev), _t(k, dev), 3)
    ridx, rd2 = oracle.three_nn(q, k)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)
    pts = rng.random((2, 3, N1), dtype=np.float32)
    keys = np.ascontiguousarray(pts[:, :, rng.integers(0, N1, size=N2)])       # duplicated keys
    try:
        F.set_distance_mode("fmad")
        idx, d2 = F.search_nn_distance(_t(pts, dev), _t(keys, dev), 3)
    finally:
        F.set_distance_mode("strict")
    ridx, rd2 = oracle.three_nn(pts, keys, fmad=1)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)
    i32, w = F.three_nn_weights(_t(pts, dev), _t(keys, dev))
    ridx, rd2 = oracle.three_nn(pts, keys)
    assert np.array_equal(i32.cpu().numpy().astype(np.int64), ridx)
    assert np.array_equal(w.cpu().numpy(), oracle.interp_weights(rd2))


@pytest.mark.parametrize("C,N2,N1", [(1024, 256, 1024), (512, 5120, 25600), (5, 9, 100)])
def test_three_interpolate_matches_oracle(F, oracle, dev, C, N2, N1):
    rng = np.random.default_rng(C)
    feat = rng.standard_normal((2, C, N2)).astype(np.float32)
    idx = rng.integers(0, N2, size=(2, N1, 3))
    d2 = rng.random((2, N1, 3), dtype=np.float32) * np.float32(1e-3)
    d2[0, 0] = 0
    w = F.interp_weights(_t(d2, dev))
    rw = oracle.interp_weights(d2)
    assert np.array_equal(w.cpu().numpy(), rw)
    # torch's three elementwise device ops (modules.py:118-120) agree to fp32 rounding
    # (torch's GPU division / reduction are not bit-identical to its CPU ones)
    inv = 1.0 / torch.clamp(_t(d2, dev), min=1e-10)
    assert torch.allclose(inv / torch.sum(inv, dim=2, keepdim=True), w, rtol=1e-6, atol=0)
    out = F.feature_interpolate(_t(feat, dev), _t(idx, dev), w)
    assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(feat, idx, rw))


@pytest.mark.parametrize("mode", ["tile", "lane"])
@pytest.mark.parametrize("C,N2,N1", [(4, 7, 63), (68, 300, 1001), (132, 50, 4098), (512, 1024, 5120)])
def test_three_interpolate_channels_last_kernels(F, oracle, dev, monkeypatch, mode, C, N2, N1):
    """Both channels-last forms (64 x 64 tile through LDS; lane per point), ragged tiles:
    C not a multiple of the 64-channel slice, N1 not a multiple of 4 / 64; FMAD mode too."""
    monkeypatch.setenv("S4G_INTERP_MODE", mode)
    rng = np.random.default_rng(C + N1)
    feat = rng.standard_normal((3, C, N2)).astype(np.float32)
    idx = rng.integers(0, N2, size=(3, N1, 3))
    w = rng.random((3, N1, 3), dtype=np.float32)
    out = F.feature_interpolate(_t(feat, dev), _t(idx, dev), _t(w, dev))
    assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(feat, idx, w))
    try:
        F.set_distance_mode("fmad")
        out = F.feature_interpolate(_t(feat, dev), _t(idx, dev), _t(w, dev))
    finally:
        F.set_distance_mode("strict")
    assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(feat, idx, w, fmad=1))


def test_three_interpolate_backward(F, oracle, dev):
    rng = np.random.default_rng(4)
    feat = torch.from_numpy(rng.standard_normal((2, 6, 20)).astype(np.float32)).to(dev)
    feat.requires_grad_(True)
    idx = rng.integers(0, 20, size=(2, 31, 3))
    w = rng.random((2, 31, 3), dtype=np.float32)
    g = rng.standard_normal((2, 6, 31)).astype(np.float32)
    out = F.feature_interpolate(feat, _t(idx, dev), _t(w, dev))
    out.backward(_t(g, dev))
    ref = oracle.three_interpolate_backward(g, idx, w, 20)
    assert np.allclose(feat.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_gather_knn_like_reference_self_test(F, dev):
    """The reference's only test (functions/gather_knn.py:27-56): gather_knn against
    torch.gather on an expanded view, forward and backward, B=2 C=4 N=5 k=3, seed 1
    -- here with asserts instead of prints."""
    torch.manual_seed(1)
    B, N, C, k = 2, 5, 4, 3
    feat = torch.rand(B, C, N).to(dev)
    knn = torch.randint(0, N, [B, N, k]).long().to(dev)
    a = feat.clone().requires_grad_(True)
    b = feat.clone().requires_grad_(True)
    ref = torch.gather(a.unsqueeze(2).expand(B, C, N, N), 3, knn.unsqueeze(1).expand(B, C, N, k))
    got = F.gather_knn(b, knn)
    assert torch.equal(ref, got)
    ref.backward(torch.ones_like(ref))
    got.backward(torch.ones_like(got))
    assert torch.allclose(a.grad, b.grad)


def test_fmad_mode_all_geometry_ops(F, oracle, dev, bq_mode):
    """S4G_FLAG_FMAD (nvcc-style contraction) against the oracle's fmad restatement."""
    pts = synth.make_batch([6, 7], 9000, variant="dup-heavy")
    M = 1500
    try:
        F.set_distance_mode("fmad")
        idx = F.farthest_point_sample(_t(pts, dev), M).cpu().numpy()
        ref_idx = oracle.fps(pts, M, fmad=1)
        assert np.array_equal(idx, ref_idx)
        ctr = oracle.gather_points(pts, ref_idx)
        for mode in ("grid", "scan"):
            bq_mode(mode)
            bi, bc = F.ball_query(_t(pts, dev), _t(ctr, dev), 0.03, 32)
            ri, rc = oracle.ball_query(pts, ctr, 0.03, 32, fmad=1)
            assert np.array_equal(bi.cpu().numpy(), ri) and np.array_equal(bc.cpu().numpy(), rc)
        ni, nd = F.search_nn_distance(_t(pts, dev), _t(ctr, dev), 3)
        rni, rnd = oracle.three_nn(pts, ctr, fmad=1)
        assert np.array_equal(ni.cpu().numpy(), rni) and np.array_equal(nd.cpu().numpy(), rnd)
        feat = np.random.default_rng(0).standard_normal((2, 8, M)).astype(np.float32)
        w = oracle.interp_weights(rnd)
        out = F.feature_interpolate(_t(feat, dev), ni, _t(w, dev))
        assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(feat, rni, w, fmad=1))
    finally:
        F.set_distance_mode("strict")


def test_fps_tie_heavy_large(F, oracle, dev):
    """25 600 points on a coarse lattice: thousands of exact distance ties per step."""
    rng = np.random.default_rng(42)
    pts = (rng.integers(0, 12, size=(2, 3, 25600)).astype(np.float32) * np.float32(0.05))
    got = F.farthest_point_sample(_t(pts, dev), 700).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 700))


def test_fps_lattice_scene_all_picks(F, oracle, dev):
    """The `lattice` table-top scene (coordinates snapped to 2^-8 m: bench.py's `mixed_batch` leg), all 5 119
    picks through the pruned kernel: most waves hold many groups with the SAME maximum, i.e. the lane-parallel
    tie resolution (FPS_TIE_PAR) decides nearly every pick; beside a tie-free scene, and in the fmad contract."""
    pts = np.concatenate([synth.make_batch([3], 25600, variant="lattice"), synth.make_batch([4], 25600)])
    got = F.farthest_point_sample(_t(pts, dev), 5120).cpu().numpy()
    assert np.array_equal(got, oracle.fps(pts, 5120))
    try:
        F.set_distance_mode("fmad")
        got = F.farthest_point_sample(_t(pts[:1], dev), 2000).cpu().numpy()
    finally:
        F.set_distance_mode("strict")
    assert np.array_equal(got, oracle.fps(pts[:1], 2000, fmad=1))


def test_runs_on_current_stream(F, oracle, dev):
    pts = synth.make_batch([1], 2048)
    s = torch.cuda.Stream(device=dev)
    x = _t(pts, dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        idx = F.farthest_point_sample(x, 128)
    s.synchronize()
    assert np.array_equal(idx.cpu().numpy(), oracle.fps(pts, 128))


# ------------------------------------------------------------------ randomized cross-checks
@pytest.mark.parametrize("seed", range(20))
def test_random_geometry_pipeline_matches_oracle(F, oracle, dev, seed):
    """Random sizes / radii / neighbour counts through FPS -> ball_query -> group_points ->
    3-NN -> interpolate, every stage against the oracle (bit-exact), on rotating cloud kinds.
    Sizes straddle the dispatch thresholds (grid vs scan ball query, grid vs scan 3-NN,
    vector vs scalar group kernels, AoS / channels-last variants)."""
    rng = np.random.default_rng(1000 + seed)
    variant = ["tabletop-v1", "dup-heavy", "uniform-box"][seed % 3]
    B = int(rng.integers(1, 4))
    N = int(rng.choice([777, 2048, 5000, 8192, 9001, 12000]))
    M = int(rng.integers(3, min(N, 2500)))
    K = int(rng.choice([1, 7, 16, 32, 64, 100]))
    radius = float(rng.choice([0.01, 0.03, 0.08, 0.5]))
    pts = synth.make_batch(list(range(seed, seed + B)), N, variant=variant)
    tp = _t(pts, dev)
    fps = F.farthest_point_sample(tp, M)
    rfps = oracle.fps(pts, M)
    assert np.array_equal(fps.cpu().numpy(), rfps), (N, M, variant)
    ctr = oracle.gather_points(pts, rfps)
    assert np.array_equal(F.gather_points(tp, fps).cpu().numpy(), ctr)
    idx, cnt = F.ball_query(tp, _t(ctr, dev), radius, K)
    ridx, rcnt = oracle.ball_query(pts, ctr, radius, K)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(cnt.cpu().numpy(), rcnt), (N, M, K, radius)
    grouped = F.group_points(tp, idx)
    assert np.array_equal(grouped.cpu().numpy(), oracle.group_points(pts, ridx))
    C = int(rng.choice([1, 4, 6, 20]))
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    assert np.array_equal(F.group_points(_t(feat, dev), idx).cpu().numpy(), oracle.group_points(feat, ridx))
    if M >= 3:
        nidx, nd2 = F.search_nn_distance(tp, _t(ctr, dev), 3)
        rn, rd = oracle.three_nn(pts, ctr)
        assert np.array_equal(nidx.cpu().numpy(), rn) and np.array_equal(nd2.cpu().numpy(), rd), (N, M, variant)
        w = F.interp_weights(nd2)
        sparse = rng.standard_normal((B, C, M)).astype(np.float32)
        got = F.feature_interpolate(_t(sparse, dev), nidx, w).cpu().numpy()
        ref = oracle.three_interpolate(sparse, rn, w.cpu().numpy())
        assert np.array_equal(got, ref)
