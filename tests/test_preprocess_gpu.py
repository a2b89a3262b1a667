"""GPU parity tests of the pre-processing kernels (row f3) against oracle/preprocess.py:
indices / masks bit-exact, voxel means bit-exact (double accumulation in point order)."""
import numpy as np
import pytest
import torch

from oracle import preprocess as OP

pytestmark = pytest.mark.gpu


def _scene(n, seed=0, variant="tabletop-v1"):
    from s4g_release_amd import synth
    return synth.make_batch([seed], n, variant=variant)[0]          # (3, n) f32


@pytest.mark.parametrize("n", [0, 1, 63, 1024, 1025, 48902])
def test_crop_matches_oracle(dev, n):
    from s4g_release_amd import preprocess as PP
    p = _scene(max(n, 1), 2)[:, :n]
    p = np.ascontiguousarray(p)
    full = _scene(4096, 2)
    lo, hi = np.percentile(full, 10, axis=1), np.percentile(full, 85, axis=1)
    ws = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
    got = PP.filter_work_space(torch.from_numpy(p).to(dev), ws).cpu().numpy()
    ref = OP.filter_work_space(p, ws)
    assert np.array_equal(got, ref)
    if n >= 1024:
        assert 0 < len(ref) < n


def test_crop_points_on_the_bounds_are_dropped(dev):
    from s4g_release_amd import preprocess as PP
    p = np.array([[0.0, 0.4, -0.4, 0.1], [0.0, 0.0, 0.0, 0.2], [0.8, 0.8, 0.8, 0.749]], dtype=np.float32)
    got = PP.filter_work_space(torch.from_numpy(p).to(dev), PP.WORKSPACE).cpu().tolist()
    assert got == OP.filter_work_space(p, PP.WORKSPACE).tolist() == [0]


@pytest.mark.parametrize("n,voxel", [(1, 0.005), (500, 0.005), (25600, 0.005), (48902, 0.004), (20000, 0.05)])
def test_voxel_down_sample_matches_oracle(dev, n, voxel):
    from s4g_release_amd import preprocess as PP
    p = _scene(n, 5)
    got = PP.voxel_down_sample(torch.from_numpy(p).to(dev), voxel).cpu().numpy()
    ref = OP.voxel_down_sample(p, voxel)
    assert got.shape == ref.shape and got.shape[1] <= n
    assert np.array_equal(got, ref)


def test_voxel_dup_heavy_cloud(dev):
    from s4g_release_amd import preprocess as PP
    p = _scene(6000, 1, variant="dup-heavy")
    got = PP.voxel_down_sample(torch.from_numpy(p).to(dev), 0.01).cpu().numpy()
    assert np.array_equal(got, OP.voxel_down_sample(p, 0.01))


@pytest.mark.parametrize("n,nb,radius", [(1, 0, 0.02), (300, 3, 0.05), (12000, 32, 0.02), (12000, 8, 0.01),
                                         (20000, 32, 0.3)])
def test_radius_outlier_mask_matches_oracle(dev, n, nb, radius):
    from s4g_release_amd import preprocess as PP
    p = _scene(n, 7)
    got = PP.radius_outlier_mask(torch.from_numpy(p).to(dev), nb, radius).cpu().numpy()
    ref = OP.remove_radius_outlier(p, nb, radius)
    assert np.array_equal(got, ref)
    if n >= 12000 and radius <= 0.02:
        assert 0 < ref.sum() < n          # the case actually separates points


def test_radius_outlier_far_outliers_and_wide_scene(dev):
    """Isolated points far outside the 32-cell torus must not pick up aliased neighbours."""
    from s4g_release_amd import preprocess as PP
    p = _scene(8000, 3).copy()
    p[:, :5] += np.array([[3.0], [-7.0], [11.0]], dtype=np.float32)       # lone fliers
    got = PP.radius_outlier_mask(torch.from_numpy(p).to(dev), 4, 0.02).cpu().numpy()
    ref = OP.remove_radius_outlier(p, 4, 0.02)
    assert np.array_equal(got, ref) and not got[:5].any()


def test_pre_processing_pipeline_matches_oracle(dev):
    from s4g_release_amd import preprocess as PP
    cloud = _scene(30000, 11)
    pts, proc = PP.pre_processing(torch.from_numpy(cloud).to(dev), num_input=4096, seed=5)
    ref = OP.pre_processing(cloud, PP.VOXEL_SIZE, PP.NUM_POINTS_THRESHOLD, PP.RADIUS_THRESHOLD, 4096, 5)
    assert pts.shape == (3, 4096)
    assert np.array_equal(pts.cpu().numpy(), ref)


def test_sample_single_cloud_fps_mode(dev):
    from s4g_release_amd import preprocess as PP
    from s4g_release_amd import functions as F
    p = torch.from_numpy(_scene(5000, 13)).to(dev)
    out = PP.sample_single_cloud(p, 1024, mode="fps")
    idx = F.farthest_point_sample(p.unsqueeze(0), 1024)[0]
    assert torch.equal(out, p[:, idx])
    short = PP.sample_single_cloud(p[:, :100], 256, seed=2, mode="fps")   # falls back to repetition
    assert short.shape == (3, 256)


@pytest.mark.parametrize("seed", range(10))
def test_random_preprocessing_matches_oracle(dev, seed):
    """Random cloud kinds / sizes / voxel edges / radii / thresholds through crop, voxel
    down-sample and radius-outlier removal, each against the oracle (bit-exact)."""
    from s4g_release_amd import preprocess as PP
    rng = np.random.default_rng(500 + seed)
    variant = ["tabletop-v1", "dup-heavy", "uniform-box"][seed % 3]
    n = int(rng.choice([37, 1000, 4097, 9000, 15000]))
    p = _scene(n, seed, variant=variant)
    t = torch.from_numpy(p).to(dev)
    lo, hi = np.percentile(p, rng.uniform(0, 30), axis=1), np.percentile(p, rng.uniform(60, 100), axis=1)
    ws = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
    assert np.array_equal(PP.filter_work_space(t, ws).cpu().numpy(), OP.filter_work_space(p, ws))
    voxel = float(rng.choice([0.002, 0.005, 0.013, 0.04]))
    got = PP.voxel_down_sample(t, voxel).cpu().numpy()
    ref = OP.voxel_down_sample(p, voxel)
    assert got.shape == ref.shape and np.array_equal(got, ref), (n, voxel, variant)
    radius = float(rng.choice([0.008, 0.02, 0.05]))
    nb = int(rng.choice([1, 4, 16, 32]))
    gm = PP.radius_outlier_mask(torch.from_numpy(ref).to(dev), nb, radius).cpu().numpy()
    assert np.array_equal(gm, OP.remove_radius_outlier(ref, nb, radius)), (n, nb, radius, variant)


def test_crop_equals_the_reference(dev):
    """`filter_work_space` against the mask `CloudPreProcessor.filter_work_space`
    (cloud_processor/cloud_processor.py:12-27) itself produced (tools/gen_golden_post.py)."""
    import os
    from s4g_release_amd import preprocess as PP
    px = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_crop.npz"))
    got = PP.filter_work_space(torch.from_numpy(px["cloud"]).to(dev), tuple(px["workspace"])).cpu().numpy()
    assert np.array_equal(got, np.nonzero(px["valid"])[0])
    assert 0 < got.size < px["cloud"].shape[1]
