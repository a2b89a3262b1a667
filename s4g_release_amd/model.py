"""The S4G network (`MODEL.TYPE: "PN2_CLS"`): 3 set-abstraction layers, 3
feature-propagation layers and four per-point heads.

Mirror of reference `network_models/models/PointNet2_tcls.py:10-153` (class
`PointNet2`) with the hyper-parameters of
`grasp_proposal/configs/curvature_model.yaml:11-22` as defaults of
`S4GConfig`.  The module tree reproduces the reference's 200 state_dict
entries, so `load_checkpoint` accepts a real
`curvature_model.pth` (`{"model": state_dict}`, optional `module.` prefixes --
reference `utils/checkpoint.py:31,54-55,81-88`).  Loss / metric classes are
training-only and out of scope.
"""
from dataclasses import dataclass, field
from typing import Tuple

import torch
from torch import nn

from .modules import PointNetSAModule, PointnetFPModule
from .nn_utils import SharedMLP


@dataclass
class S4GConfig:
    """Values of curvature_model.yaml:11-22 (+ yacs defaults DATA.SCORE_CLASSES=3,
    NUM_REMOVAL_DIRECTIONS=5)."""
    score_classes: int = 3
    num_input: int = 25600
    num_centroids: Tuple[int, ...] = (5120, 1024, 256)
    radius: Tuple[float, ...] = (0.02, 0.08, 0.32)
    num_neighbours: Tuple[int, ...] = (64, 64, 64)
    sa_channels: Tuple[Tuple[int, ...], ...] = ((128, 128, 256), (256, 256, 512), (512, 512, 1024))
    fp_channels: Tuple[Tuple[int, ...], ...] = ((1024, 1024), (512, 512), (256, 256, 256))
    num_fp_neighbours: Tuple[int, ...] = (3, 3, 3)
    seg_channels: Tuple[int, ...] = (512, 256, 256, 128)
    num_removal_directions: int = 5
    dropout_prob: float = 0.5

    def model_kwargs(self):
        return dict(score_classes=self.score_classes, num_centroids=self.num_centroids,
                    radius=self.radius, num_neighbours=self.num_neighbours,
                    sa_channels=self.sa_channels, fp_channels=self.fp_channels,
                    num_fp_neighbours=self.num_fp_neighbours, seg_channels=self.seg_channels,
                    num_removal_directions=self.num_removal_directions,
                    dropout_prob=self.dropout_prob)


class PointNet2(nn.Module):
    """PointNet++ single-scale-grouping backbone with S4G's four heads.

    forward({"scene_points": (B,3,N) f32}) ->
      {"score": (B,score_classes,N), "frame_R": (B,9,N), "frame_t": (B,4,N),
       "movable_logits": (B,num_removal_directions,N)  (after Sigmoid)}
    """
    _SA_MODULE = PointNetSAModule
    _FP_MODULE = PointnetFPModule

    def __init__(self, score_classes, num_centroids=(10240, 1024, 128, 0),
                 radius=(0.2, 0.3, 0.4, -1.0), num_neighbours=(64, 64, 64, -1),
                 sa_channels=((32, 32, 64), (64, 64, 128), (128, 128, 256), (256, 512, 1024)),
                 fp_channels=((256, 256), (256, 128), (128, 128), (64, 64, 64)),
                 num_fp_neighbours=(0, 3, 3, 3), seg_channels=(128,), num_removal_directions=5,
                 dropout_prob=0.5):
        super().__init__()
        n_sa, n_fp = len(num_centroids), len(fp_channels)
        assert len(radius) == n_sa and len(num_neighbours) == n_sa and len(sa_channels) == n_sa
        assert n_sa == n_fp and len(num_fp_neighbours) == n_fp

        # channel bookkeeping as PointNet2_tcls.py:56-80
        self.sa_modules = nn.ModuleList()
        c = 0
        for i in range(n_sa):
            self.sa_modules.append(self._SA_MODULE(in_channels=c, mlp_channels=sa_channels[i],
                                                   num_centroids=num_centroids[i], radius=radius[i],
                                                   num_neighbours=num_neighbours[i], use_xyz=True))
            c = sa_channels[i][-1]
        skip = [0] + [ch[-1] for ch in sa_channels]
        self.fp_modules = nn.ModuleList()
        c = skip[-1]
        for i in range(n_fp):
            self.fp_modules.append(self._FP_MODULE(in_channels=c + skip[-2 - i],
                                                   mlp_channels=fp_channels[i],
                                                   num_neighbors=num_fp_neighbours[i]))
            c = fp_channels[i][-1]

        # heads (PointNet2_tcls.py:83-95)
        self.mlp_seg = SharedMLP(c, seg_channels, ndim=1, dropout_prob=dropout_prob)
        self.seg_logit = nn.Conv1d(seg_channels[-1], score_classes, 1, bias=True)
        self.mlp_R = SharedMLP(c, seg_channels, ndim=1)
        self.R_logit = nn.Conv1d(seg_channels[-1], 9, 1, bias=True)
        self.mlp_t = SharedMLP(c, seg_channels, ndim=1)
        self.t_logit = nn.Conv1d(seg_channels[-1], 4, 1, bias=True)
        self.mlp_movable = SharedMLP(c, seg_channels, ndim=1, dropout_prob=dropout_prob)
        self.movable_logit = nn.Sequential(
            nn.Conv1d(seg_channels[-1], num_removal_directions, 1, bias=True), nn.Sigmoid())

    def forward(self, data_batch):
        xyz = data_batch["scene_points"]
        feature = None
        level_xyz, level_feature = [xyz], [feature]
        for sa in self.sa_modules:
            xyz, feature = sa(xyz, feature)
            level_xyz.append(xyz)
            level_feature.append(feature)
        sparse_xyz, sparse_feature = xyz, feature
        for i, fp in enumerate(self.fp_modules):
            dense_xyz, dense_feature = level_xyz[-2 - i], level_feature[-2 - i]
            sparse_feature = fp(dense_xyz, sparse_xyz, dense_feature, sparse_feature)
            sparse_xyz = dense_xyz
        x = sparse_feature
        return {"score": self.seg_logit(self.mlp_seg(x)),
                "frame_R": self.R_logit(self.mlp_R(x)),
                "frame_t": self.t_logit(self.mlp_t(x)),
                "movable_logits": self.movable_logit(self.mlp_movable(x))}


def build_pointnet2_cls(cfg=None):
    """Counterpart of PointNet2_tcls.py:270-283 for an `S4GConfig`."""
    cfg = cfg or S4GConfig()
    return PointNet2(**cfg.model_kwargs())


def randomize_bn_(model, seed):
    """Give every BatchNorm non-trivial affine parameters and running statistics.

    Default init (gamma=1, beta=0, mean=0, var=1) would make BN folding
    untestable; golden fixtures and parity tests call this after seeding
    (SURVEY.md section 8c "Weights")."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 1.0 + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.0 + 0.5)
    return model


def strip_module_prefix(state_dict):
    """DataParallel checkpoints prefix keys with `module.` (utils/checkpoint.py:81-88)."""
    if all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return state_dict


def load_checkpoint(model, path, strict=True):
    """Load a reference-format checkpoint: torch.save({"model": state_dict, ...})."""
    blob = torch.load(path, map_location="cpu")
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    model.load_state_dict(strip_module_prefix(sd), strict=strict)
    return model


def calibrate_bn_(model, seed, data_batch, decades=1.0, beta_std=0.5):
    """BatchNorm parameters of a TRAINED-LIKE network: gamma log-uniform over `decades` around 1,
    beta = gamma N(0, beta_std), and running statistics CALIBRATED on the network's own activations --
    one pass over `data_batch` with only the BatchNorm layers in train mode and momentum 1, so that
    `running_mean` / `running_var` are what the preceding layers really produce (as after training).

    `randomize_bn_`'s sigma^2 in [0.5, 1.5] sits orders of magnitude above the real variance of a
    default-initialised network's activations: the signal dies in the first layers and every output is
    a per-channel constant -- no good as a parity fixture.  With calibrated statistics each layer is
    re-normalised to unit variance, so the outputs depend on the input at every point.

    Works on any module tree with BatchNorm1d/2d (the product's `PointNet2` and the reference's, through
    whichever operators that network calls).  `SharedMLP` stays in eval mode: no dropout."""
    g = torch.Generator().manual_seed(seed)
    bns = [m for m in model.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
    with torch.no_grad():
        for m in bns:
            C = m.weight.numel()
            gamma = 10.0 ** ((torch.rand(C, generator=g) - 0.5) * decades)
            m.weight.copy_(gamma)
            m.bias.copy_(gamma * torch.randn(C, generator=g) * beta_std)
        model.eval()
        saved = [m.momentum for m in bns]
        for m in bns:
            m.train()
            m.momentum = 1.0
        model(data_batch)
        for m, mom in zip(bns, saved):
            m.eval()
            m.momentum = mom
    return model


def bn_state(state_dict):
    """The BatchNorm entries of a state_dict (everything a calibrated fixture stores by value; the
    convolutions are regenerated from the seed)."""
    return {k: v for k, v in state_dict.items() if ".bn." in k}
