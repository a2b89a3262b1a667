"""Shared-MLP building blocks with the reference's module names and state_dict
layout (reference `network_models/nn_utils/conv.py:6-83`, `mlp.py:55-114`,
`init.py:4-8`): per layer `conv` (1x1, bias only without BN) -> `bn`
(eps 1e-5, momentum 0.1) -> ReLU.  A real `curvature_model.pth` therefore loads
unchanged.

These modules are the *reference-shaped* path (plain torch ops on whatever
device the tensors live on).  The fast inference path folds BN into the
weights and runs the contraction on MFMA (`fused.py`).
"""
import torch.nn.functional as F
from torch import nn


def init_bn(module):
    """gamma = 1, beta = 0 (reference nn_utils/init.py:4-8)."""
    if module.weight is not None:
        nn.init.ones_(module.weight)
    if module.bias is not None:
        nn.init.zeros_(module.bias)


class _ConvNd(nn.Module):
    _conv = None
    _bn = None

    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1,
                 **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.conv = self._conv(in_channels, out_channels, kernel_size, bias=(not bn), **kwargs)
        self.bn = self._bn(out_channels, momentum=bn_momentum) if bn else None
        self.relu = nn.ReLU(inplace=True) if relu else None
        self.init_weights()

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.relu is not None:
            x = self.relu(x)
        return x

    def init_weights(self, init_fn=None):
        if init_fn is not None:
            init_fn(self.conv)
        if self.bn is not None:
            init_bn(self.bn)


class Conv1d(_ConvNd):
    """reference nn_utils/conv.py:6-43"""
    _conv = nn.Conv1d
    _bn = nn.BatchNorm1d


class Conv2d(_ConvNd):
    """reference nn_utils/conv.py:46-83"""
    _conv = nn.Conv2d
    _bn = nn.BatchNorm2d


class SharedMLP(nn.ModuleList):
    """Per-position MLP shared over 1 or 2 trailing dims (reference nn_utils/mlp.py:55-114).

    Children are indexed `0..L-1`, each a Conv1d/Conv2d block, so parameter
    names are `<prefix>.<i>.conv.weight`, `<prefix>.<i>.bn.*`.
    Dropout is applied only while training (mlp.py:99-105).
    """

    def __init__(self, in_channels, mlp_channels, ndim=1, dropout_prob=0.0, bn=True,
                 bn_momentum=0.1):
        super().__init__()
        if ndim not in (1, 2):
            raise ValueError("SharedMLP only supports ndim=(1, 2).")
        block = Conv1d if ndim == 1 else Conv2d
        self.in_channels = in_channels
        self.out_channels = mlp_channels[-1]
        self.ndim = ndim
        c = in_channels
        for width in mlp_channels:
            self.append(block(c, width, 1, relu=True, bn=bn, bn_momentum=bn_momentum))
            c = width
        assert dropout_prob >= 0.0
        self.dropout_prob = dropout_prob

    def forward(self, x):
        drop = F.dropout if self.ndim == 1 else F.dropout2d
        for layer in self:
            x = layer(x)
            if self.training and self.dropout_prob > 0.0:
                x = drop(x, p=self.dropout_prob, training=True)
        return x

    def init_weights(self, init_fn=None):
        for layer in self:
            layer.init_weights(init_fn)

    def extra_repr(self):
        return "dropout_prob={}".format(self.dropout_prob) if self.dropout_prob > 0.0 else ""
