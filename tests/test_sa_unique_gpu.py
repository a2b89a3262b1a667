"""The first SA level on a centroid's DISTINCT rows only (ABI 8).

ball_query pads a ball with fewer than K hits by repeating its first hit
(ball_query_kernel.cu:64-67: `if (cnt == 0) for (l < K) idx[l] = k`), group_points copies those rows
(grouping_kernel.cu:48-51) and `torch.max(new_feature, 3)` (modules.py:243) cannot see the copies: the
shared MLP only has to run on `count` rows per centroid.  s4g_group_rel_xyz_unique_i32 writes those rows
back to back (4-row granules), the chain kernel's segmented max epilogue merges a centroid's pieces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unique_reference(pts, ctr, gidx, cnt, K):
    """numpy restatement of the layout in include/s4g_ops.h (s4g_group_rel_xyz_unique_i32)."""
    B, _, N = pts.shape
    M = ctr.shape[2]
    cap = M * K
    rel = np.zeros((B * cap, 4), np.float32)
    seg4 = np.full((B * cap // 4,), -2, np.int32)           # -2: never written
    row_start = np.zeros((B, M), np.int32)
    rows = np.zeros((B,), np.int32)
    for b in range(B):
        total = sum((max(int(c), 1) + 3) // 4 * 4 for c in cnt[b])
        # a mostly-full scene keeps the plain layout: every slot, rows == M K -- and so does one whose tile-padded
        # compact row count would REACH M K (rows == M K is the plain layout's marker; only below 2 048 rows)
        dense = total > cap // 8 * 7 or (total + 255) // 256 * 256 >= cap
        r = 0
        for m in range(M):
            c4 = K if dense else (max(int(cnt[b, m]), 1) + 3) // 4 * 4
            row_start[b, m] = r
            j = gidx[b, m, :c4]
            rel[b * cap + r:b * cap + r + c4, :3] = (pts[b][:, j] - ctr[b][:, m:m + 1]).T
            seg4[(b * cap + r) // 4:(b * cap + r + c4) // 4] = b * M + m
            r += c4
        rows[b] = (r + 255) // 256 * 256
        seg4[(b * cap + r) // 4:(b * cap + rows[b]) // 4] = -1
    return rel, seg4, row_start, rows


@pytest.mark.parametrize("variant,N,M,radius", [("tabletop-v1", 4096, 512, 0.03), ("uniform-box", 2048, 256, 0.05),
                                                ("dup-heavy", 4096, 512, 0.02),
                                                ("tabletop-v1", 8192, 256, 0.06)])     # balls mostly full: plain layout
def test_group_rel_xyz_unique_layout(dev, variant, N, M, radius):
    from oracle import oracle as O
    from s4g_release_amd import _cabi, synth
    K = 64
    pts = synth.make_batch([1, 2, 3], N, variant=variant)
    idx = O.fps(pts, M)
    ctr = O.gather_points(pts, idx)
    if variant == "uniform-box":
        ctr = ctr.copy()
        ctr[0, :, 5] += 10.0                 # a centroid far from every point: an EMPTY ball (count 0)
    gidx, cnt = O.ball_query(pts, ctr, radius, K)
    assert cnt.min() < K, "the case must contain padded balls"
    dense_case = radius == 0.06
    B = pts.shape[0]
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else \
        torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    d_pts, d_ctr, d_idx, d_cnt = t(pts), t(ctr), t(gidx, torch.int32), t(cnt, torch.int32)
    rel = torch.full((B * M * K, 4), float("nan"), device=dev)
    seg4 = torch.full((B * M * K // 4,), -2, dtype=torch.int32, device=dev)
    row_start = torch.empty((B, M), dtype=torch.int32, device=dev)
    rows = torch.empty((B,), dtype=torch.int32, device=dev)
    rc = _cabi.lib().s4g_group_rel_xyz_unique_i32(d_pts.data_ptr(), d_ctr.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(),
                                                  B, N, M, K, rel.data_ptr(), seg4.data_ptr(), row_start.data_ptr(),
                                                  rows.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "group_rel_xyz_unique")
    torch.cuda.synchronize()
    r_rel, r_seg, r_start, r_rows = _unique_reference(pts, ctr, gidx, cnt, K)
    assert np.array_equal(rows.cpu().numpy(), r_rows)
    assert (r_rows == M * K).any() == dense_case, r_rows
    assert np.array_equal(row_start.cpu().numpy(), r_start)
    assert np.array_equal(seg4.cpu().numpy(), r_seg)
    got = rel.cpu().numpy()
    for b in range(B):      # every row a scene occupies, bit for bit; nothing behind it is touched
        lo, hi = b * M * K, b * M * K + r_rows[b]
        assert np.array_equal(got[lo:hi], r_rel[lo:hi])
        assert np.isnan(got[hi:(b + 1) * M * K]).all()
    if variant == "uniform-box":
        assert cnt[0, 5] == 0 and r_start[0, 6] - r_start[0, 5] == 4     # the empty ball keeps four copies of point 0


@pytest.mark.parametrize("M,counts", [(16, [64] * 12 + [32] * 4),      # 896 rows = 7/8 M K, padded = 1 024 = M K
                                      (16, [64] * 11 + [32] * 4 + [60]),   # 892 rows
                                      (4, [64, 64, 64, 17]),          # M K = 256: one tile, always plain
                                      (4, [1, 1, 1, 1]),
                                      (16, [20] * 16),                # 320 rows -> 512: compact stays legal
                                      (32, [64] * 28 + [1] * 4)])     # 2 048 rows: 1 808 > 7/8: plain by the old rule
def test_small_levels_never_mix_the_plain_marker_with_compact_rows(dev, M, counts):
    """below 2 048 rows a compact scene's padded row count could equal M K -- the plain
    layout's marker -- while its rows sat at compact offsets, so the contraction read centroid m at row m K and
    found another centroid's rows.  Either layout must be self-consistent: rows == M K <=> row_start[m] == m K."""
    from s4g_release_amd import _cabi
    K, N, B = 64, 500, 2
    rng = np.random.default_rng(M * 1000 + len(counts) + sum(counts))
    pts = rng.standard_normal((B, 3, N)).astype(np.float32)
    ctr = rng.standard_normal((B, 3, M)).astype(np.float32)
    cnt = np.tile(np.asarray(counts, np.int64), (B, 1))
    cnt[1] = cnt[1][::-1]
    gidx = rng.integers(0, N, (B, M, K))
    for b in range(B):
        for m in range(M):
            gidx[b, m, cnt[b, m]:] = gidx[b, m, 0]          # ball_query's padding
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    d_pts, d_ctr, d_idx, d_cnt = t(pts, torch.float32), t(ctr, torch.float32), t(gidx, torch.int32), t(cnt, torch.int32)
    rel = torch.full((B * M * K, 4), float("nan"), device=dev)
    seg4 = torch.full((B * M * K // 4,), -2, dtype=torch.int32, device=dev)
    row_start = torch.empty((B, M), dtype=torch.int32, device=dev)
    rows = torch.empty((B,), dtype=torch.int32, device=dev)
    rc = _cabi.lib().s4g_group_rel_xyz_unique_i32(d_pts.data_ptr(), d_ctr.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(),
                                                  B, N, M, K, rel.data_ptr(), seg4.data_ptr(), row_start.data_ptr(),
                                                  rows.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "group_rel_xyz_unique")
    torch.cuda.synchronize()
    r_rel, r_seg, r_start, r_rows = _unique_reference(pts, ctr, gidx, cnt, K)
    h_rows, h_start = rows.cpu().numpy(), row_start.cpu().numpy()
    assert np.array_equal(h_rows, r_rows) and np.array_equal(h_start, r_start)
    for b in range(B):
        plain = h_rows[b] == M * K
        assert plain == np.array_equal(h_start[b], np.arange(M) * K)
        lo, hi = b * M * K, b * M * K + r_rows[b]
        assert np.array_equal(rel.cpu().numpy()[lo:hi], r_rel[lo:hi])
        assert np.array_equal(seg4.cpu().numpy()[lo // 4:hi // 4], r_seg[lo // 4:hi // 4])


def test_model_with_a_tiny_first_level_matches_the_full_rows_model(dev, monkeypatch):
    """The same hazard end to end: a first SA level of 16 centroids x 64 neighbours (M K = 1 024)."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import PointNet2, randomize_bn_
    torch.manual_seed(31)
    net = PointNet2(score_classes=3, num_centroids=(16, 8, 4), radius=(0.16, 0.3, 0.6), num_neighbours=(64, 64, 64),
                    sa_channels=((128, 128, 256), (256, 256, 512), (512, 512, 1024)),
                    fp_channels=((1024, 1024), (512, 512), (256, 256, 256)), num_fp_neighbours=(3, 3, 3),
                    seg_channels=(512, 256, 256, 128), num_removal_directions=5, dropout_prob=0.5)
    net = randomize_bn_(net, 32).to(dev).eval()
    pts = torch.from_numpy(synth.make_batch(list(range(12)), 2048)).to(dev)
    a, ia = FusedPointNet2(net)({"scene_points": pts}, return_intermediates=True)
    rows = ia.pop("sa0_rows", None)
    monkeypatch.setenv("S4G_SA_UNIQUE", "0")
    b, ib = FusedPointNet2(net)({"scene_points": pts}, return_intermediates=True)
    for k in ia:
        assert torch.equal(ia[k], ib[k]), k
    for k in a:
        err = (a[k] - b[k]).abs().max().item()
        assert err < 2e-5 * max(1.0, b[k].abs().max().item()), (k, err, None if rows is None else rows.tolist())


def test_unsupported_shapes_are_refused(dev):
    from s4g_release_amd import _cabi
    z = torch.zeros(4096, device=dev)
    zi = torch.zeros(4096, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = _cabi.lib().s4g_group_rel_xyz_unique_i32
    a = (z.data_ptr(), z.data_ptr(), zi.data_ptr(), zi.data_ptr())
    o = (z.data_ptr(), zi.data_ptr(), zi.data_ptr(), zi.data_ptr(), st)
    assert f(*a, 1, 64, 4, 30, *o) == _cabi.S4G_EUNSUPPORTED      # K % 4
    assert f(*a, 1, 64, 3, 16, *o) == _cabi.S4G_EUNSUPPORTED      # M K % 256


def _net(dev, seed=3):
    from tests import golden_util as GU
    return GU.shipped_net(dev)


@pytest.mark.parametrize("precision", ["f16x2", "bf16"])
def test_model_on_distinct_rows_equals_the_full_rows_model(dev, monkeypatch, precision):
    """Same indices, same maxima; the hidden layer's per-tile scales see other rows, so f16x2 agrees to
    fp32 round-off (not bitwise); the scale-free bf16 form is bit-identical."""
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    net = _net(dev)
    crowded = synth.make_batch([8], 25600) * np.float32(0.55)       # the same cloud shrunk: balls mostly full
    pts = torch.from_numpy(np.concatenate([synth.make_batch([5, 6], 25600),
                                           synth.make_batch([7], 25600, variant="uniform-box"), crowded])).to(dev)
    a, ia = FusedPointNet2(net, precision=precision)({"scene_points": pts}, return_intermediates=True)
    assert int(ia["cnt0"].min()) < 64      # padded balls exist (the uniform scene is almost all padding)
    rows = ia.pop("sa0_rows").cpu().numpy()
    # per scene: the two table-top scenes and the sparse one are compacted, the crowded one keeps the plain layout
    assert (rows[:3] < 0.875 * 5120 * 64).all() and rows[3] == 5120 * 64 and rows[2] < 0.15 * 5120 * 64
    monkeypatch.setenv("S4G_SA_UNIQUE", "0")
    b, ib = FusedPointNet2(net, precision=precision)({"scene_points": pts}, return_intermediates=True)
    for k in ia:
        assert torch.equal(ia[k], ib[k]), k
    for k in a:
        assert torch.equal(a[k][3], b[k][3]), k      # the plain-layout scene runs the very same arithmetic
        if precision == "bf16":
            assert torch.equal(a[k], b[k]), k
        else:
            err = (a[k] - b[k]).abs().max().item()
            assert err < 2e-5 * max(1.0, b[k].abs().max().item()), (k, err)


def test_distinct_rows_model_matches_the_oracle_and_is_batch_invariant(dev):
    from oracle import pn2_forward
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    from s4g_release_amd.model import S4GConfig
    cfg = S4GConfig()
    net = _net(dev, 11)
    pts = synth.make_batch([2, 9, 4], 25600)
    runner = FusedPointNet2(net)
    assert runner.sa_unique
    got = runner({"scene_points": torch.from_numpy(pts).to(dev)})
    alone = runner({"scene_points": torch.from_numpy(pts[1:2]).to(dev)})
    for k in got:
        assert torch.equal(got[k][1:2], alone[k]), k          # a scene's layout never depends on its batch
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = pn2_forward.forward(sd, pts[1:2], cfg.num_centroids, cfg.radius, cfg.num_neighbours)
    from tests.ref64 import forward64
    ref64 = forward64(sd, pts[1:2], cfg.num_centroids, cfg.radius, cfg.num_neighbours)
    for k, v in ref.items():      # calibrated weights: within 1e-4 of scale of float64, and of torch's fp32 by the triangle
        scale = max(1.0, float(np.abs(ref64[k]).max()))
        got_k = alone[k].cpu().numpy().astype(np.float64)
        o64 = float(np.abs(v - ref64[k]).max()) / scale
        assert float(np.abs(got_k - ref64[k]).max()) / scale < 1e-4, k
        assert float(np.abs(got_k - v).max()) / scale < 1e-4 + o64, k
