"""CPU tests of the pre-processing oracle (row f3) on hand-checkable clouds."""
import numpy as np

from oracle import preprocess as OP


def test_crop_is_strict_and_ordered():
    p = np.array([[0.0, 0.4, -0.4, 0.1, 0.39999],
                  [0.0, 0.0, 0.0, 0.2, 0.0],
                  [0.8, 0.8, 0.8, 0.749, 1.1]], dtype=np.float32)
    ws = [-0.40, 0.40, -0.4, 0.4, 0.749, 1.2]
    # x = 0.4 / -0.4 sit ON the bound (float32(0.4) == float32 of the list value) -> dropped;
    # z = 0.749 is on the lower bound -> dropped
    assert OP.filter_work_space(p, ws).tolist() == [0, 4]


def test_voxel_means_and_order():
    # two points in one 1-cm voxel, one in the next x cell, one a z layer above
    p = np.array([[0.001, 0.003, 0.012, 0.002],
                  [0.001, 0.002, 0.001, 0.001],
                  [0.001, 0.001, 0.001, 0.013]], dtype=np.float32)
    out = OP.voxel_down_sample(p, 0.01)
    # origin = min - 0.005 -> cells (0,0,0) x2, (1,0,0), (0,0,1); ascending (iz,iy,ix)
    assert out.shape == (3, 3)
    np.testing.assert_allclose(out[:, 0], [(0.001 + 0.003) / 2, 0.0015, 0.001], rtol=1e-6)
    np.testing.assert_allclose(out[:, 1], [0.012, 0.001, 0.001], rtol=1e-6)
    np.testing.assert_allclose(out[:, 2], [0.002, 0.001, 0.013], rtol=1e-6)
    origin, dims = OP.voxel_grid(p, 0.01)
    assert dims.tolist() == [2, 1, 2]


def test_radius_outlier_counts_include_self_and_are_strict():
    # 4 points on a line, spacing 1; r = 1 -> strict '<' sees only the point itself
    p = np.zeros((3, 4), dtype=np.float32)
    p[0] = [0, 1, 2, 3]
    assert OP.radius_neighbour_counts(p, 1.0).tolist() == [1, 1, 1, 1]
    assert OP.radius_neighbour_counts(p, 1.5).tolist() == [2, 3, 3, 2]
    assert OP.remove_radius_outlier(p, 2, 1.5).tolist() == [False, True, True, False]


def test_sample_indices_permutation_and_repetition():
    a = OP.sample_indices(1000, 256, seed=3)
    assert len(set(a.tolist())) == 256 and a.max() < 1000
    assert np.array_equal(a, OP.sample_indices(1000, 256, seed=3))
    assert not np.array_equal(a, OP.sample_indices(1000, 256, seed=4))
    b = OP.sample_indices(100, 256, seed=3)
    assert len(b) == 256 and sorted(set(b.tolist())) == list(range(100))
    assert np.array_equal(b[:100], b[100:200])


def test_host_sampler_matches_oracle():
    from s4g_release_amd import preprocess as PP
    for n, k, seed in ((1000, 256, 3), (100, 256, 9), (25600, 25600, 0)):
        assert np.array_equal(PP.sample_indices(n, k, seed), OP.sample_indices(n, k, seed))


def test_pipeline_shapes():
    rng = np.random.default_rng(0)
    cloud = (rng.random((3, 4000)).astype(np.float32) * np.array([[0.2], [0.2], [0.02]], dtype=np.float32))
    out = OP.pre_processing(cloud, 0.005, 8, 0.02, 2048, seed=1)
    assert out.shape == (3, 2048) and out.dtype == np.float32
