"""PointNet++ set-abstraction / feature-propagation modules over the MI355X
operator API -- drop-in for the five classes `PN2_CLS` instantiates from the
reference's `pointnet2_utils/modules.py`:

  FarthestPointSampler  :9-27      QueryGrouper       :30-57
  FeatureInterpolator   :96-132    PointNetSAModule   :174-250
  PointnetFPModule      :477-510

Same constructor arguments, forward signatures, return values and (empty)
parameter sets.  The Avg/MSG/Edge* variants are out of scope (not instantiated
by either shipped yaml).
"""
import torch
from torch import nn

from . import functions as _F
from .nn_utils import SharedMLP


class FarthestPointSampler(nn.Module):
    def __init__(self, num_centroids):
        super().__init__()
        self.num_centroids = num_centroids

    def forward(self, points):
        with torch.no_grad():
            return _F.farthest_point_sample(points, self.num_centroids)

    def extra_repr(self):
        return "num_centroids={:d}".format(self.num_centroids)


class QueryGrouper(nn.Module):
    def __init__(self, radius, num_neighbours):
        super().__init__()
        assert radius > 0.0 and num_neighbours > 0
        self.radius = radius
        self.num_neighbours = num_neighbours

    def forward(self, new_xyz, xyz, feature, use_xyz):
        if torch.is_grad_enabled() and xyz.requires_grad:
            with torch.no_grad():
                index, _ = _F.ball_query(xyz, new_xyz, self.radius, self.num_neighbours)
            group_xyz = _F.group_points(xyz, index)       # (B, 3, M, K), fresh tensor, differentiable
        else:
            # the same two operators as ONE pass over the outputs (s4g_query_group_f32: identical
            # index / grouped tensors, 13 % less time than the two launches)
            with torch.no_grad():
                index, _, group_xyz = _F.query_and_group(xyz, new_xyz, self.radius, self.num_neighbours)
        group_xyz -= new_xyz.unsqueeze(-1)                # centroid-relative (modules.py:44)
        if feature is None:
            return group_xyz, group_xyz
        group_feature = _F.group_points(feature, index)   # (B, C, M, K)
        if use_xyz:
            group_feature = torch.cat([group_xyz, group_feature], dim=1)   # [xyz, feat] (:50)
        return group_feature, group_xyz

    def extra_repr(self):
        return "radius={}, num_neighbours={}".format(self.radius, self.num_neighbours)


class FeatureInterpolator(nn.Module):
    def __init__(self, num_neighbors, eps=1e-10):
        super().__init__()
        self.num_neighbors = num_neighbors
        self._eps = eps

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature):
        with torch.no_grad():
            index, distance = _F.search_nn_distance(dense_xyz, sparse_xyz, self.num_neighbors)
            # w = 1/max(d2, eps), normalised over the 3 neighbours (modules.py:118-120)
            weight = _F.interp_weights(distance, self._eps)
        interpolated = _F.feature_interpolate(sparse_feature, index, weight)
        if dense_feature is None:
            return interpolated
        return torch.cat([interpolated, dense_feature], dim=1)   # [interp, skip] (:125)

    def extra_repr(self):
        return "num_neighbours={:d}, eps={}".format(self.num_neighbors, self._eps)


class PointNetSAModule(nn.Module):
    """Sample (FPS) -> group (ball query) -> shared MLP -> max over neighbours."""

    def __init__(self, in_channels, mlp_channels, num_centroids, radius, num_neighbours, use_xyz):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = mlp_channels[-1]
        self.num_centroids = num_centroids
        self.use_xyz = use_xyz
        self.mlp = SharedMLP(in_channels + (3 if use_xyz else 0), mlp_channels, ndim=2, bn=True)
        self.sampler = FarthestPointSampler(num_centroids) if num_centroids > 0 else None
        if num_neighbours < 0:
            assert radius < 0.0
            self.grouper = None
        else:
            assert num_neighbours > 0 and radius > 0.0
            self.grouper = QueryGrouper(radius, num_neighbours)

    def forward(self, xyz, feature=None):
        if self.num_centroids == 0:
            # one group holding every point, centred at the origin (modules.py:224-231)
            assert self.grouper is None
            new_xyz = xyz.new_zeros(xyz.size(0), 3, 1)
            group_feature = feature.unsqueeze(2)
            if self.use_xyz:
                group_feature = torch.cat([xyz.unsqueeze(2), group_feature], dim=1)
        else:
            if self.num_centroids == -1:
                new_xyz = xyz
            else:
                new_xyz = _F.gather_points(xyz, self.sampler(xyz))
            group_feature, _ = self.grouper(new_xyz, xyz, feature, use_xyz=self.use_xyz)
        new_feature = self.mlp(group_feature)
        new_feature, _ = torch.max(new_feature, 3)
        return new_xyz, new_feature

    def init_weights(self, init_fn=None):
        self.mlp.init_weights(init_fn)

    def extra_repr(self):
        return "num_centroids={:d}, use_xyz={}".format(self.num_centroids, self.use_xyz)


class PointnetFPModule(nn.Module):
    """3-NN inverse-distance interpolation -> concat skip -> shared MLP."""

    def __init__(self, in_channels, mlp_channels, num_neighbors):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = mlp_channels[-1]
        self.mlp = SharedMLP(in_channels, mlp_channels, ndim=1, bn=True)
        if num_neighbors == 0:
            self.interpolator = None
        elif num_neighbors == 3:
            self.interpolator = FeatureInterpolator(num_neighbors)
        else:
            raise ValueError("Expected value 1 or 3, but {} given.".format(num_neighbors))

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature):
        if self.interpolator is None:
            assert sparse_xyz.size(2) == 1 and sparse_feature.size(2) == 1
            expanded = sparse_feature.expand(-1, -1, dense_xyz.size(2))
            new_feature = torch.cat([expanded, dense_feature], dim=1)
        else:
            new_feature = self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature)
        return self.mlp(new_feature)

    def init_weights(self, init_fn=None):
        self.mlp.init_weights(init_fn)
