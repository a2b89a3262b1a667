"""`detector.GraspDetector`: the reference's `GraspDetector.detect` (grasp_detector.py:187-254) composed on the device --
subsample + REAL2TRAIN, FusedPointNet2(topk=K), thresholds / decode / Gram-Schmidt / frame, batched collision check against
the whole input cloud, survivor compaction, importance sampling -- against the CPU oracle's restatement of each stage
(oracle/postprocess.py, oracle/preprocess.py; the stages themselves are pinned by fixtures the reference's own functions
produced: tests/test_post_golden.py, tests/test_postprocess_gpu.py).  The network is the calibrated one
(tests/golden_util.shipped_net): its expected scores have a real order over a scene."""
import numpy as np
import pytest
import torch

from tests import golden_util as GU

pytestmark = pytest.mark.gpu
K = 2048


def _clouds(n, ids):
    """Raw clouds in the REAL frame: tabletop scenes mapped through TRAIN2REAL (the inverse of what the detector applies)."""
    from s4g_release_amd import synth
    t = synth.make_batch(ids, n)                                     # TRAIN frame (B, 3, n)
    return np.ascontiguousarray(np.stack([t[:, 1], t[:, 0], -t[:, 2]], axis=1))


def _threshold(scores, rank=600):
    """A score threshold in a clear gap of the sorted expected scores near `rank` (so that fp32 vs float64 rounding of a
    borderline score cannot change the candidate set)."""
    s = np.sort(scores)[::-1]
    gaps = s[rank - 50:rank + 50] - s[rank - 49:rank + 51]
    j = int(np.argmax(gaps)) + rank - 50
    assert s[j] - s[j + 1] > 1e-5
    return float(0.5 * (s[j] + s[j + 1]))


@pytest.mark.parametrize("n", [25600, 40000])
def test_detect_equals_the_oracle_stage_by_stage(dev, n):
    from oracle import postprocess as OP, preprocess as OPre
    from s4g_release_amd import postprocess as PP
    from s4g_release_amd.detector import GraspDetector
    from s4g_release_amd.fused import FusedPointNet2
    net = GU.shipped_net(dev)
    cam = np.eye(4)
    cam[:3, :3] = np.linalg.qr(np.random.default_rng(5).standard_normal((3, 3)))[0]
    det = GraspDetector(net, topk=K, camera2base=cam, seed=3)
    cloud = _clouds(n, [6, 7])
    d_cloud = torch.from_numpy(cloud).to(dev)
    # stage 1: the subsample the reference's `sample_single_cloud` would draw with these (seeded) indices, REAL2TRAIN
    pts = det.pre_processing(d_cloud).cpu().numpy()
    r2t = np.asarray(PP.REAL2TRAIN, dtype=np.float32)
    for b in range(2):
        idx = OPre.sample_indices(n, 25600, 3 + b)
        want = (r2t[:3, :3] @ cloud[b][:, idx]).astype(np.float32)
        assert np.array_equal(pts[b], want), b
    # the full forward's outputs (pinned elsewhere) feed the oracle's stages
    full = FusedPointNet2(net)({"scene_points": torch.from_numpy(pts).to(dev)})
    pred = {k: v.cpu().numpy() for k, v in full.items()}
    es = [OP.expected_scores(pred["score"][b]) for b in range(2)]
    thr = max(_threshold(es[b]) for b in range(2))
    vthr = -0.2
    u = np.random.default_rng(9).random(5)
    out = det.detect_device(d_cloud, num_selected=5, score_threshold=thr, verticalness_threshold=vthr, uniforms=u)
    poses, scores, count = [t.cpu().numpy() for t in out]
    H, s, index, cnt = [t.cpu().numpy() for t in out.candidates]
    dm = cam[:3, :3] @ np.asarray(OP.TRAIN2REAL)[:3, :3]
    # the candidates BEFORE the collision check, recomputed by the standalone stage (itself checked against the oracle)
    pre = PP.detect_poses(det.run({"scene_points": torch.from_numpy(pts).to(dev)}, topk=K), torch.from_numpy(pts).to(dev),
                          thr, vthr, direction_matrix=dm, frame=PP.TRAIN2REAL, max_poses=K)
    ok_dev, _ = PP.view_non_collision(pre[0], d_cloud, inverse="se3")
    for b in range(2):
        rH, rs, ridx = OP.detector_post_processing({k: v[b] for k, v in pred.items()}, pts[b], thr, vthr, dm)
        m = int(pre[3][b])
        assert m == len(ridx) and 100 < m < K
        assert np.array_equal(pre[2][b, :m].cpu().numpy(), ridx)                       # same candidates, best first
        assert np.allclose(pre[0][b, :m].cpu().numpy(), rH, atol=3e-5)
        # stage 4: collision verdicts against the WHOLE cloud (n points, REAL frame), analytic SE(3) inverse
        rok, _ = OP.view_non_collision(rH[None].astype(np.float32), cloud[b:b + 1],
                                       global2local=OP.se3_inverse_f32(rH)[None])
        okb = ok_dev[b, :m].cpu().numpy()
        assert (okb == rok[0]).mean() >= 0.95                      # (a count at its threshold may differ by an ulp-close point)
        keep = np.nonzero(okb)[0]
        c = int(cnt[b])
        assert c == len(keep) and np.array_equal(index[b, :c], ridx[keep])             # survivors, order kept
        assert np.allclose(H[b, :c], rH[keep], atol=3e-5) and np.allclose(s[b, :c], rs[keep], atol=2e-6)
        assert (index[b, c:] == -1).all() and (H[b, c:] == 0).all()
        # stage 5: importance sampling with the same draws
        if c > 5:
            pick = OP.importance_sampling(s[b, :c].astype(np.float64), u)
            assert count[b] == 5 and np.allclose(poses[b], H[b][pick], atol=0) and np.allclose(scores[b], s[b][pick])
        else:
            assert count[b] == c and np.array_equal(poses[b, :c], H[b, :c])


def test_batch_graph_mask_and_reference_signature(dev):
    from s4g_release_amd.detector import GraspDetector
    net = GU.shipped_net(dev)
    det = GraspDetector(net, topk=K, seed=1)
    cloud = _clouds(30000, [2, 3, 4])
    d = torch.from_numpy(cloud).to(dev)
    kw = dict(num_selected=5, score_threshold=0.6, verticalness_threshold=-2.0)
    full = det.detect_device(d, **kw)
    cand = [t.clone() for t in full.candidates]
    torch.cuda.synchronize()
    assert int(cand[3].min()) > 5
    # a scene's detections do not depend on its batch (seed b of the batch = seed + b alone)
    for b in range(3):
        one = GraspDetector(det.run, topk=K, seed=1 + b).detect_device(d[b:b + 1].contiguous(), **kw)
        for x, y in zip(one.candidates, cand):
            assert torch.equal(x[0], y[b])
    # one HIP graph for the whole call: same candidates, picks drawn inside the graph stay in range
    g = det.graph(d, **kw)
    for rep in range(2):
        out = g(d)
        torch.cuda.synchronize()
        for x, y in zip(out.candidates, cand):
            assert torch.equal(x, y)
        assert (out[2] == 5).all() and float(out[1].min()) > 0.6
    with pytest.raises(RuntimeError):
        g(d[:1])
    # the reference's signature: (n, 3) numpy in, trimmed (poses, scores) out; the mask restricts what the NETWORK sees
    # (grasp_detector.py:196-199), the collision check still sees the whole cloud (:220-222)
    mask = np.zeros(30000, dtype=bool)
    mask[:28000] = True
    poses, scores = det.detect(cloud[0].T.copy(), cloud_mask=mask, **kw)
    assert poses.shape == (5, 4, 4) and scores.shape == (5,)
    sub = GraspDetector(det.run, topk=K, seed=1).detect_device(
        torch.from_numpy(np.ascontiguousarray(cloud[:1, :, :28000])).to(dev), collision_cloud=d[:1], **kw)
    allc = det.detect_device(torch.from_numpy(np.ascontiguousarray(cloud[:1, :, :28000])).to(dev), **kw)
    torch.cuda.synchronize()
    assert int(sub.candidates[3]) <= int(allc.candidates[3])          # 2 000 more obstacle points can only remove grasps
    with pytest.raises(AssertionError):
        det.detect(np.zeros((5, 4), np.float32))
    # stage timers (the reference logs the same five stages: :203,:209,:231,:251)
    det.stage_events = []
    det.detect_device(d, **kw)
    ms = det.stage_ms()
    det.stage_events = None
    assert list(ms) == ["pre_processing", "prediction", "post_processing", "collision_check", "importance_sampling"]
    assert all(v >= 0 for v in ms.values())


def test_intended_preprocessing_mode_runs_the_voxel_and_outlier_passes(dev):
    from s4g_release_amd import preprocess as pre
    from s4g_release_amd.detector import GraspDetector
    net = GU.shipped_net(dev)
    cloud = torch.from_numpy(_clouds(40000, [8])).to(dev)
    det = GraspDetector(net, topk=K, seed=2, preprocess="intended")
    pts = det.pre_processing(cloud)
    want, kept = pre.pre_processing(cloud[0], 25600, 2)
    assert torch.equal(pts[0], want) and kept.shape[1] < 40000       # voxel + outlier passes did shrink the cloud
    out = det.detect_device(cloud, score_threshold=0.6, verticalness_threshold=-2.0)
    torch.cuda.synchronize()
    assert int(out.candidates[3]) > 0


def test_collision_counts_over_padded_lists_skip_the_padding_rows(dev):
    """ABI 12 `s4g_collision_counts_n_f32`: with device-side counts the first count[b] rows get exactly the plain entry
    point's counts and verdicts, the padding rows zero counts and ok = False."""
    from s4g_release_amd import postprocess as PP, synth
    rng = np.random.default_rng(4)
    B, N, Kp = 3, 20000, 64
    pts = torch.from_numpy(synth.make_batch([1, 2, 3], N)).to(dev)
    pred = {k: torch.from_numpy(rng.standard_normal((B, c, N)).astype(np.float32)).to(dev)
            for k, c in (("score", 3), ("frame_R", 9), ("frame_t", 4))}
    H, _, _ = PP.decode_top_poses(pred, pts, Kp)
    count = torch.tensor([64, 10, 0], device=dev)
    ok_all, c_all = PP.view_non_collision(H, pts, inverse="se3")
    ok_n, c_n = PP.view_non_collision(H, pts, inverse="se3", count=count)
    for b, n in enumerate(count.tolist()):
        assert torch.equal(c_n[b, :n], c_all[b, :n]) and torch.equal(ok_n[b, :n], ok_all[b, :n])
        assert (c_n[b, n:] == 0).all() and not ok_n[b, n:].any()
