"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the cloud pre-processing
in front of the network ("next" row f3).  Parity unpinned: the reference delegates
voxelisation and outlier removal to open3d (>= 0.12, requirements.txt:2; not
installed here) and discards their results (cloud_processor.py:34,40), and its
subsampling is unseeded (grasp_detector.py:82-92).  What is restated:

  filter_work_space   cloud_processor.py:12-29 (strict inequalities, order kept)
  voxel_down_sample   open3d geometry::PointCloud::VoxelDownSample: origin =
                      min_bound - voxel/2, index = floor((p - origin)/voxel),
                      output = mean of the voxel's points.  Output order is
                      unspecified in open3d (unordered_map); here ascending
                      (iz, iy, ix).  Mean = double sum in point order, one rounding.
  remove_radius_outlier  open3d RemoveRadiusOutliers: keep p iff the radius search
                      around p (p included, squared distance < r^2 as in nanoflann's
                      RadiusResultSet) returns more than nb_points points.  Distances
                      in the canonical fp32 arithmetic ((dx*dx + dy*dy) + dz*dz).
  pre_processing      grasp_detector.py:94-105: voxelize, remove outliers, REAL2TRAIN
                      transform (:26), subsample to num_input (seeded here).
"""
import numpy as np

REAL2TRAIN = np.array([[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float32)


def filter_work_space(points_3n, workspace):
    """-> ascending indices of the points strictly inside the box."""
    p = np.asarray(points_3n, dtype=np.float32)
    w = np.asarray(workspace, dtype=np.float32)
    keep = (p[0] > w[0]) & (p[0] < w[1]) & (p[1] > w[2]) & (p[1] < w[3]) & (p[2] > w[4]) & (p[2] < w[5])
    return np.nonzero(keep)[0].astype(np.int32)


def voxel_grid(points_3n, voxel):
    """(origin (3,) f32, dims (3,) int32) of the voxel grid of a cloud."""
    p = np.asarray(points_3n, dtype=np.float32)
    v = np.float32(voxel)
    origin = (p.min(axis=1) - v * np.float32(0.5)).astype(np.float32)
    top = np.floor(((p.max(axis=1) - origin) / v).astype(np.float32)).astype(np.int64)
    return origin, (top + 1).astype(np.int32)


def voxel_down_sample(points_3n, voxel):
    p = np.asarray(points_3n, dtype=np.float32)
    if p.shape[1] == 0:
        return p.copy()
    v = np.float32(voxel)
    origin, dims = voxel_grid(p, voxel)
    idx = np.floor(((p - origin[:, None]).astype(np.float32) / v).astype(np.float32)).astype(np.int64)
    idx = np.clip(idx, 0, dims.astype(np.int64)[:, None] - 1)
    key = (idx[2] * int(dims[1]) + idx[1]) * int(dims[0]) + idx[0]
    order = np.argsort(key, kind="stable")               # point order inside a voxel
    ks = key[order]
    heads = np.nonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))[0]
    sums = np.add.reduceat(p[:, order].astype(np.float64), heads, axis=1)
    # reduceat on float64 adds left to right within a segment for these sizes? not guaranteed:
    # recompute sequentially to pin the order
    out = np.empty((3, len(heads)), dtype=np.float32)
    ends = np.concatenate([heads[1:], [len(ks)]])
    for v_i, (a, b) in enumerate(zip(heads, ends)):
        s = np.zeros(3, dtype=np.float64)
        for j in order[a:b]:
            s += p[:, j].astype(np.float64)
        out[:, v_i] = (s / float(b - a)).astype(np.float32)
    del sums
    return out


def radius_neighbour_counts(points_3n, radius, chunk=512):
    """Number of points (self included) at canonical fp32 squared distance < r^2."""
    p = np.asarray(points_3n, dtype=np.float32)
    n = p.shape[1]
    r2 = np.float32(radius) * np.float32(radius)
    cnt = np.zeros(n, dtype=np.int64)
    for a in range(0, n, chunk):
        q = p[:, a:a + chunk]
        dx = (p[0][None, :] - q[0][:, None]).astype(np.float32)
        dy = (p[1][None, :] - q[1][:, None]).astype(np.float32)
        dz = (p[2][None, :] - q[2][:, None]).astype(np.float32)
        d = ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32)
        d = (d + (dz * dz).astype(np.float32)).astype(np.float32)
        cnt[a:a + chunk] = (d < r2).sum(axis=1)
    return cnt


def remove_radius_outlier(points_3n, nb_points, radius):
    """-> boolean keep mask."""
    return radius_neighbour_counts(points_3n, radius) > int(nb_points)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
        return z ^ (z >> np.uint64(31))


def sample_indices(n, num_input, seed):
    """Seeded stand-in for grasp_detector.py:82-92 (np.random.choice without /
    with replacement): a splitmix64-keyed permutation, repeated when n < num_input."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        keys = _splitmix64(np.arange(n, dtype=np.uint64) + base)
    perm = np.argsort(keys, kind="stable")
    if n >= num_input:
        return perm[:num_input]
    reps = (num_input + n - 1) // n
    return np.tile(perm, reps)[:num_input]


def pre_processing(cloud_3n, voxel, nb_points, radius, num_input, seed):
    p = voxel_down_sample(cloud_3n, voxel)
    p = p[:, remove_radius_outlier(p, nb_points, radius)]
    p = (REAL2TRAIN[:3, :3] @ p + REAL2TRAIN[:3, 3:4]).astype(np.float32)
    return p[:, sample_indices(p.shape[1], num_input, seed)]
