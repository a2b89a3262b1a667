"""tools/run_scene.py -- the counterpart of the reference's demo harness
(`grasp_proposal/grasp_proposal_test.py:17-86`): cloud file -> seeded subsample -> forward ->
synchronised timing -> outputs.  Runs it as a program on the reference's own sample scene (the
seeded 25 600-point subsample kept as a data fixture) and checks the written predictions against
the golden outputs captured from the reference's Python network."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import golden_util as GU

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_scene.py")] + args,
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_run_scene_on_the_reference_sample_scene_matches_the_golden_outputs(tmp_path):
    """Default weights = the calibrated golden run's network: the harness's written predictions against what the
    REFERENCE's network produced for this scene (tests/golden/pn2_calib_full.npz, scene "real"), 1e-4 of scale."""
    g = GU.load("pn2_calib_full.npz")
    cloud = tmp_path / "scene.npy"
    np.save(cloud, GU.calib_scenes(g)["real"][0])        # (3, 25600): the seeded subsample of the reference's sample scene
    pred = tmp_path / "pred.npz"
    rep = _run([str(cloud), "--reps", "3", "--out", str(pred), "--topk", "10"], tmp_path)
    assert rep["points"] == 25600 and rep["source_points"] == 25600 and rep["precision"] == "f16x2"
    assert rep["weights"] == "calibrated"
    assert rep["forward_ms"]["p10"] <= rep["forward_ms"]["median"] <= rep["forward_ms"]["p90"]
    assert rep["scenes_per_sec"] > 10 and rep["outputs"]["frame_R"] == [1, 9, 25600]
    z = np.load(pred)
    GU.calib_compare_full(g, "real", {k: z[k] for k in GU.HEADS}, {}, need_levels=())
    assert z["pose_H"].shape == (1, 10, 4, 4) and (np.diff(z["pose_score"][0]) <= 0).all()
    assert float(np.ptp(z["pose_score"][0])) > 1e-4      # the decoded scores have an order (a flat score head would tie)


def test_run_scene_seeded_weights_still_match_the_index_pinning_fixture(tmp_path):
    g = GU.load("pn2_real.npz")
    cloud = tmp_path / "scene.npy"
    np.save(cloud, g["points"][0])
    pred = tmp_path / "pred.npz"
    rep = _run([str(cloud), "--reps", "2", "--out", str(pred), "--weights", "seeded"], tmp_path)
    assert rep["weights"].startswith("seeded random")
    z = np.load(pred)
    pos = g["positions"]
    for k in ("score", "frame_R", "frame_t", "movable_logits"):
        assert np.max(np.abs(z[k][:, :, pos] - g["out/" + k])) < 1e-4, k


def test_run_scene_subsamples_larger_and_smaller_clouds_reproducibly(tmp_path):
    a = _run(["synthetic:3", "--reps", "2", "--out", str(tmp_path / "a.npz")], tmp_path)       # 48 902 -> 25 600
    b = _run(["synthetic:3", "--reps", "2", "--out", str(tmp_path / "b.npz")], tmp_path)
    assert a["source_points"] == 48902 and a["points"] == 25600
    za, zb = np.load(tmp_path / "a.npz"), np.load(tmp_path / "b.npz")
    assert np.array_equal(za["points"], zb["points"]) and np.array_equal(za["score"], zb["score"])
    small = tmp_path / "small.npy"
    np.save(small, np.load(tmp_path / "a.npz")["points"][0][:, :9000].T)      # (N, 3) layout, N < 25 600
    c = _run([str(small), "--reps", "2"], tmp_path)
    assert c["source_points"] == 9000 and c["points"] == 25600                # drawn with replacement (:28-29)
