"""Cloud pre-processing on the HIP device ("next" row f3): the steps in front of the
network in the reference's `GraspDetector._pre_processing`
(grasp_proposal/grasp_detector.py:94-105) and `CloudPreProcessor`
(grasp_proposal/cloud_processor/cloud_processor.py:12-42).

Differences a maintainer should know:
  * the reference calls open3d's `voxel_down_sample` / `remove_radius_outlier` but
    drops the clouds they return (cloud_processor.py:34,40), so its network input is
    the raw cloud; here both steps APPLY (`CloudPreProcessor.points` is updated),
    which is what the config constants (processing_config.py:20-22) describe;
  * open3d emits voxels in hash-map order; here they come out in ascending
    (iz, iy, ix) cell order;
  * subsampling (`sample_single_cloud`, grasp_detector.py:82-92) is seeded
    (splitmix64-keyed permutation) or, as the reference's comment suggests, FPS.
Every step is a HIP kernel behind the C ABI (include/s4g_ops.h); there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _cabi
from . import functions as _F

# processing_config.py:17-23
TABLE_HEIGHT = 0.75
WORKSPACE = (-0.40, 0.40, -0.4, 0.4, TABLE_HEIGHT - 0.001, TABLE_HEIGHT + 0.45)
VOXEL_SIZE = 0.005
NUM_POINTS_THRESHOLD = 32
RADIUS_THRESHOLD = 0.02
# grasp_detector.py:26
REAL2TRAIN = ((0., 1., 0., 0.), (1., 0., 0., 0.), (0., 0., -1., 0.), (0., 0., 0., 1.))


def _cloud(points):
    p = _F._f32c(points, "points")
    if p.dim() != 2 or p.size(0) != 3:
        raise RuntimeError("points must be (3, N)")
    return p


def filter_work_space(points, workspace=WORKSPACE):
    """Indices (ascending, int64) of the points strictly inside the workspace box."""
    p = _cloud(points)
    n = p.size(1)
    index = torch.empty(max(n, 1), dtype=torch.int32, device=p.device)
    count = torch.zeros(1, dtype=torch.int32, device=p.device)
    ws = (ctypes.c_float * 6)(*[float(v) for v in workspace])
    with torch.cuda.device(p.device):
        rc = _cabi.lib().s4g_crop_indices_f32(_F._ptr(p), n, ws, index.data_ptr(), count.data_ptr(),
                                              _F._stream())
    _cabi.check(rc, "crop_indices")
    return index[:int(count.item())].long()


def voxel_down_sample(points, voxel_size=VOXEL_SIZE):
    """(3, V) voxel means (open3d VoxelDownSample semantics, cells in ascending order)."""
    p = _cloud(points)
    n = p.size(1)
    if n == 0:
        return p.clone()
    v = np.float32(voxel_size)
    lo = p.min(dim=1)[0].cpu().numpy()
    hi = p.max(dim=1)[0].cpu().numpy()
    origin = (lo - v * np.float32(0.5)).astype(np.float32)
    dims = (np.floor(((hi - origin) / v).astype(np.float32)).astype(np.int64) + 1)
    if int(dims[0]) * int(dims[1]) * int(dims[2]) >= 2 ** 32:
        raise RuntimeError("voxel grid too fine for this cloud (more than 2^32 cells)")
    out = torch.empty((3, n), dtype=torch.float32, device=p.device)
    count = torch.zeros(1, dtype=torch.int32, device=p.device)
    nbytes = _cabi.lib().s4g_voxel_down_sample_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    o3 = (ctypes.c_float * 3)(*[float(x) for x in origin])
    d3 = (ctypes.c_int32 * 3)(*[int(x) for x in dims])
    with torch.cuda.device(p.device):
        rc = _cabi.lib().s4g_voxel_down_sample_f32(p.data_ptr(), n, float(v), o3, d3, out.data_ptr(),
                                                   count.data_ptr(), ws.data_ptr(), nbytes,
                                                   _F._stream())
    _cabi.check(rc, "voxel_down_sample")
    return out[:, :int(count.item())].contiguous()


def radius_outlier_mask(points, nb_points=NUM_POINTS_THRESHOLD, radius=RADIUS_THRESHOLD):
    """Boolean keep mask of open3d's RemoveRadiusOutliers (more than nb_points points,
    the point itself included, within the radius)."""
    p = _cloud(points)
    n = p.size(1)
    keep = torch.zeros(n, dtype=torch.uint8, device=p.device)
    if n == 0:
        return keep.bool()
    nbytes = _cabi.lib().s4g_radius_outlier_workspace_bytes(n)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=p.device)
    with torch.cuda.device(p.device):
        rc = _cabi.lib().s4g_radius_outlier_mask_f32(p.data_ptr(), n, float(radius), int(nb_points),
                                                     keep.data_ptr(), ws.data_ptr(), nbytes,
                                                     _F._DIST_FLAGS, _F._stream())
    _cabi.check(rc, "radius_outlier_mask")
    return keep.bool()


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
        return z ^ (z >> np.uint64(31))


def sample_indices(n, num_input, seed=0):
    """Seeded counterpart of `np.random.choice(n, num_input, replace=n < num_input)`
    (grasp_detector.py:86-89): a keyed permutation, repeated when the cloud is short."""
    if n <= 0:
        raise RuntimeError("cannot sample from an empty cloud")
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        keys = _splitmix64(np.arange(n, dtype=np.uint64) + base)
    perm = np.argsort(keys, kind="stable")
    if n >= num_input:
        return perm[:num_input]
    return np.tile(perm, (num_input + n - 1) // n)[:num_input]


class CloudPreProcessor:
    """Device counterpart of the reference class of the same name; `points` is (3, N)."""

    def __init__(self, cloud):
        self.points = _cloud(cloud)

    def filter_work_space(self, workspace=WORKSPACE):
        valid_index = filter_work_space(self.points, workspace)
        self.points = self.points[:, valid_index].contiguous()
        return valid_index

    def voxelize(self, voxel_size=VOXEL_SIZE):
        self.points = voxel_down_sample(self.points, voxel_size)

    def remove_outliers(self, nb_points=NUM_POINTS_THRESHOLD, radius=RADIUS_THRESHOLD):
        self.points = self.points[:, radius_outlier_mask(self.points, nb_points, radius)].contiguous()


def sample_single_cloud(points, num_input=25600, seed=0, mode="random"):
    """(3, N) -> (3, num_input).  mode "random": seeded permutation / repetition;
    "fps": farthest point sampling when N > num_input (the strategy the reference's
    comment at grasp_detector.py:84 proposes), same kernel as the network's sampler."""
    p = _cloud(points)
    n = p.size(1)
    if mode == "fps" and n > num_input:
        idx = _F.farthest_point_sample(p.unsqueeze(0), num_input)[0]
    elif mode in ("random", "fps"):
        idx = torch.from_numpy(sample_indices(n, num_input, seed).astype(np.int64)).to(p.device)
    else:
        raise ValueError("mode must be 'random' or 'fps'")
    return p[:, idx].contiguous()


def pre_processing(cloud, num_input=25600, seed=0, mode="random", workspace=None):
    """`GraspDetector._pre_processing`: (optional crop), voxelize, remove outliers,
    REAL2TRAIN transform, subsample.  Returns (points (3, num_input), processed cloud (3, M))."""
    cp = CloudPreProcessor(cloud)
    if workspace is not None:
        cp.filter_work_space(workspace)
    cp.voxelize()
    cp.remove_outliers()
    t = torch.tensor(REAL2TRAIN, dtype=torch.float32, device=cp.points.device)
    pts = (t[:3, :3] @ cp.points + t[:3, 3:4]).contiguous()
    return sample_single_cloud(pts, num_input, seed, mode), cp.points
