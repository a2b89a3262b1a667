"""The heads for a serving path that only decodes a scene's best-scoring points (ABI 11, `s4g_heads_desc_t.head_mask`;
`FusedPointNet2(..., topk=K)`): the score head on every point, the rotation / translation / movable heads on the K kept
points.  Checked against the full forward: the same points are kept in the same order, their 21 channels agree to fp32
round-off (the hidden layers' per-tile power-of-two scales see other rows), the decoded grasp frames are the full
path's -- `GraspDetector.post_processing` (grasp_detector.py:137-185) and the demo's top-K decode
(file_logger_cls.py:196-218) only ever read those points."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(dev, seed=21):
    """The shipped architecture with the calibrated golden run's weights (tests/golden_util.shipped_net): its score head
    has a real order over a scene's points (a `randomize_bn_` network's is flat to 1e-7 -- every point ties)."""
    from tests import golden_util as GU
    return GU.shipped_net(dev)


@pytest.mark.parametrize("precision,K", [("f16x2", 2048), ("f16x2", 100), ("bf16", 1024)])
def test_kept_points_carry_the_full_forwards_outputs(dev, precision, K):
    from s4g_release_amd import postprocess as pp, synth
    from s4g_release_amd.fused import FusedPointNet2, PackedPred
    run = FusedPointNet2(_net(dev), precision=precision)
    pts = torch.from_numpy(synth.make_batch([4, 5, 6], 25600 - 24)).to(dev)      # a ragged last panel in both launches
    full = run({"scene_points": pts})
    kept = run({"scene_points": pts}, topk=K)
    assert isinstance(kept, PackedPred) and kept.packed.shape == (3, 21, K) and kept["index"].shape == (3, K)
    es = pp.expected_score(full["score"].contiguous())
    assert float(es.std()) > 1e-3, "the test network's scores must vary over a scene"
    ref_sel = torch.topk(es, K, dim=1, largest=True, sorted=True)[1]
    idx = kept["index"]
    # the score head ran on every point with the very same arithmetic: the kept SET is the full forward's top K
    # (order may differ only between points whose expected scores tie to the last bit)
    assert torch.equal(torch.sort(idx, dim=1)[0], torch.sort(ref_sel, dim=1)[0])
    tol = 2e-5 if precision == "f16x2" else 3e-2
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        want = torch.gather(full[k], 2, idx.unsqueeze(1).expand(-1, full[k].shape[1], -1))
        err = (kept[k] - want).abs().max().item()
        assert err < tol * max(1.0, want.abs().max().item()), (k, err)
    assert torch.equal(kept["score"], torch.gather(full["score"], 2, idx.unsqueeze(1).expand(-1, 3, -1)))   # pass 1 IS the full score head


def test_decoded_frames_equal_the_full_paths(dev):
    from s4g_release_amd import postprocess as pp, synth
    from s4g_release_amd.fused import FusedPointNet2
    run = FusedPointNet2(_net(dev, 31))
    pts = torch.from_numpy(synth.make_batch([7, 8], 25600)).to(dev)
    full = run({"scene_points": pts})
    kept = run({"scene_points": pts}, topk=2048)
    Hf, sf, jf = pp.decode_top_poses(full, pts, 50)
    Hk, sk, jk = pp.decode_top_poses(kept, pts, 50)
    assert torch.equal(jf, jk) and torch.equal(sf, sk)
    assert (Hf - Hk).abs().max().item() < 2e-5
    # the detector's post-processing: thresholds chosen so that candidates exist and fewer than K pass the score test
    es = pp.expected_score(full["score"].contiguous(), "detector")
    thr = float(torch.sort(es, dim=1, descending=True)[0][:, 700].max())       # <= 700 candidates per scene: fewer than K
    a = pp.detect_poses(full, pts, score_threshold=thr, verticalness_threshold=-2.0, max_poses=256)
    b = pp.detect_poses(kept, pts, score_threshold=thr, verticalness_threshold=-2.0, max_poses=256)
    assert int(a[3].min()) > 0 and torch.equal(a[3], b[3]) and torch.equal(a[2], b[2])
    assert (a[0] - b[0]).abs().max().item() < 2e-5 and torch.equal(a[1], b[1])
    with pytest.raises(ValueError):
        pp.detect_poses(kept, pts, reference_indexing=True)


def test_pipelined_submissions_with_kept_points(dev):
    from s4g_release_amd import synth
    from s4g_release_amd.fused import FusedPointNet2
    run = FusedPointNet2(_net(dev, 41))
    batches = [torch.from_numpy(synth.make_batch([i, i + 1], 25600)).to(dev) for i in (0, 2, 4)]
    seq = [run({"scene_points": b}, topk=512) for b in batches]
    seq = [{k: v.clone() for k, v in p.items()} for p in seq]
    hs = [run.submit({"scene_points": b}, topk=512) for b in batches]
    for want, h in zip(seq, hs):
        got = h.result()
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(want[k], got[k]), k


def test_head_mask_contract(dev):
    """All four heads = mask 0 = mask 15, bit for bit; a masked-out head's output pointer may be NULL and is not
    written; mask 16 is refused."""
    from s4g_release_amd import _cabi, synth
    from s4g_release_amd.fused import FusedPointNet2
    run = FusedPointNet2(_net(dev, 51))
    pts = torch.from_numpy(synth.make_batch([1], 8192)).to(dev)
    full = run({"scene_points": pts})
    calls = []
    orig = run._heads

    def spy(x, x_amax, outs, B, N0, pre=None, head_mask=0):
        calls.append(head_mask)
        if head_mask == 0:
            orig(x, x_amax, outs, B, N0, pre=pre, head_mask=15)
        else:
            orig(x, x_amax, outs, B, N0, pre=pre, head_mask=head_mask)
    run._heads = spy
    again = run({"scene_points": pts})
    assert calls == [0] and all(torch.equal(full[k], again[k]) for k in full)
    sentinel = torch.full((1, 9, 8192), 7.0, device=dev)

    def only_score(x, x_amax, outs, B, N0, pre=None, head_mask=0):
        orig(x, x_amax, [outs[0], sentinel, None, None], B, N0, pre=pre, head_mask=1)
    run._heads = only_score
    part = run({"scene_points": pts})
    torch.cuda.synchronize()
    assert torch.equal(part["score"], full["score"]) and (sentinel == 7.0).all()
    d = _cabi.HeadsDesc()
    d.precision, d.P, d.N, d.ldx = 3, 128, 64, 256
    d.C, d.H0, d.H1, d.H2, d.H3 = 256, 512, 256, 256, 128
    d.head_mask = 16
    buf = torch.zeros(1 << 16, device=dev)
    d.X = buf.data_ptr()
    for l in range(5):
        d.W_frag[l] = d.bias[l] = d.w_inv_scale[l] = buf.data_ptr()
    d.a_amax_floor = 1.0
    assert _cabi.lib().s4g_heads_chain_f32(ctypes.byref(d), None) == -1
