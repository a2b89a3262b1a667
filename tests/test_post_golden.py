"""The oracle's restatements of the steps around the network against fixtures that the
REFERENCE'S OWN functions produced (tools/gen_golden_post.py imports
grasp_proposal/grasp_detector.py, cloud_processor/view_collision_checker.py and
cloud_processor/cloud_processor.py with open3d / yacs as import-only stubs and calls
`GraspDetector.post_processing`, `orthogonalization`, `CloudCollisionChecker.view_non_collision`,
`CloudPreProcessor.filter_work_space` unbound).  SURVEY.md section 8f rows f1, f2, f3 (crop)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def det():
    return np.load(os.path.join(GOLDEN, "post_detector.npz"))


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_oracle_post_processing_literal_equals_reference(det, case):
    """grasp_detector.py:137-185 as written, incl. its two indexing quirks (:150-154)."""
    from oracle import postprocess as OP
    pred = {k: det["%s_%s" % (case, k)][0] for k in ("score", "frame_R", "frame_t")}
    sthr, vthr = det[case + "_thresholds"]
    H, s, idx = OP.detector_post_processing(pred, det[case + "_points"], sthr, vthr,
                                            det["direction_matrix"], literal=True)
    ref_H, ref_s = det[case + "_mat44"], det[case + "_scores"]
    assert H.shape == ref_H.shape and (case == "d") == (H.shape[0] == 0)
    assert np.array_equal(s, ref_s)
    assert np.array_equal(H, ref_H)              # same arithmetic in the same dtypes: bit for bit
    assert (np.diff(idx) > 0).all()              # ascending point order, NOT score order


def test_oracle_default_mode_differs_from_the_reference_as_written(det):
    """The corrected pairing (every pose from its own point's rotation) is a different function:
    same survivors of the score threshold, different rotations."""
    from oracle import postprocess as OP
    pred = {k: det["a_" + k][0] for k in ("score", "frame_R", "frame_t")}
    sthr, vthr = det["a_thresholds"]
    H, s, idx = OP.detector_post_processing(pred, det["a_points"], sthr, vthr, det["direction_matrix"])
    assert (np.diff(s) <= 0).all()
    assert H.shape[0] != det["a_mat44"].shape[0] or not np.allclose(H, det["a_mat44"])


def test_oracle_orthogonalization_equals_reference(det):
    from oracle import postprocess as OP
    assert np.array_equal(OP._orthogonalization(det["orth_rot"], det["orth_trans"]), det["orth_mat44"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_view_non_collision_equals_reference(case):
    """cloud_processor/view_collision_checker.py:37-65 on the SE(3) inverse grasp_detector.py:219
    feeds it."""
    from oracle import postprocess as OP
    cx = np.load(os.path.join(GOLDEN, "post_collision.npz"))
    g2l = cx[case + "_global2local"]
    ok, counts = OP.view_non_collision(cx[case + "_poses"][None], cx[case + "_cloud"][None],
                                       global2local=g2l[None])
    assert np.array_equal(ok[0], cx[case + "_ok"])
    assert 0 < ok.sum() < ok.size
    # the restated analytic inverse (utils/math_utils.py:26-40) is the one the reference used
    assert np.array_equal(OP.se3_inverse_f32(cx[case + "_poses"]), g2l)


def test_oracle_crop_equals_reference():
    from oracle import preprocess as OPre
    px = np.load(os.path.join(GOLDEN, "post_crop.npz"))
    idx = OPre.filter_work_space(px["cloud"], px["workspace"])
    assert np.array_equal(idx, np.nonzero(px["valid"])[0])
    assert np.array_equal(px["cloud"].T[idx].astype(np.float64), px["kept"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_oracle_demo_top_frames_equals_reference(case):
    """utils/file_logger_cls.py `loggin_to_file(with_label=False)`: expected score, translation, top-50,
    fp32 Gram-Schmidt, float64 inverse, collision filter -- the reference's own return value."""
    from oracle import postprocess as OP
    dx = np.load(os.path.join(GOLDEN, "post_demo.npz"))
    pred = {k: dx["%s_%s" % (case, k)] for k in ("score", "frame_R", "frame_t")}
    H, s, idx = OP.demo_top_frames(pred, dx[case + "_points"], K=50)
    assert 0 < H.shape[0] < 50 and H.shape == dx[case + "_top_H"].shape      # the filter cut, and not everything
    assert np.array_equal(s, dx[case + "_top_score"]) and (np.diff(s) < 0).all()
    assert np.array_equal(H, dx[case + "_top_H"])                            # bit for bit
    # the batched decode without the filter agrees on the kept frames up to its float64 Gram-Schmidt
    H2, s2, i2 = OP.decode_top_poses(pred, dx[case + "_points"], K=50)
    keep = np.isin(i2[0], idx)
    assert np.array_equal(i2[0][keep], idx) and np.array_equal(s2[0][keep], s)
    assert np.abs(H2[0][keep] - H).max() < 1e-6
