"""A training step through the reference-shaped modules on the HIP operators (the backward pair
`group_points_backward` / `interpolate_backward` behind the autograd Functions of `functions.py`, which mirror
`pointnet2_utils/functions.py:80-172`): train-mode forward (batch statistics), backward, SGD -- the loss falls, every
parameter receives a finite gradient, and with the deterministic scatters two identical steps give identical gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(score_classes=3, num_centroids=(256, 64, 16), radius=(0.08, 0.2, 0.5), num_neighbours=(16, 16, 16),
           sa_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128)), fp_channels=((128, 128), (64, 64), (32, 32, 32)),
           num_fp_neighbours=(3, 3, 3), seg_channels=(64, 32, 32, 16), num_removal_directions=5, dropout_prob=0.0)


def _setup(dev, seed=0):
    from s4g_release_amd import synth
    from s4g_release_amd.model import PointNet2
    torch.manual_seed(seed)
    net = PointNet2(**CFG).to(dev).train()
    pts = torch.from_numpy(synth.make_batch([1, 2], 2048)).to(dev)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    target = {"score": torch.randn(2, 3, 2048, generator=g).to(dev), "frame_R": torch.randn(2, 9, 2048, generator=g).to(dev),
              "frame_t": torch.randn(2, 4, 2048, generator=g).to(dev), "movable_logits": torch.rand(2, 5, 2048, generator=g).to(dev)}
    return net, pts, target


def _loss(net, pts, target):
    pred = net({"scene_points": pts})
    return sum(((pred[k] - target[k]) ** 2).mean() for k in target)


def test_sgd_steps_through_the_hip_operators_reduce_the_loss(dev):
    from s4g_release_amd import functions as F
    F.set_backward_mode("deterministic")
    net, pts, target = _setup(dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.0)
    losses = []
    for step in range(6):
        opt.zero_grad()
        loss = _loss(net, pts, target)
        loss.backward()
        if step == 0:
            for name, p in net.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), name
            nonzero = [name for name, p in net.named_parameters() if p.grad.abs().max() > 0]
            assert len(nonzero) >= 0.9 * len(list(net.parameters())), "most parameters must receive a gradient"
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and all(b < a for a, b in zip(losses, losses[1:])) and losses[-1] < 0.95 * losses[0], losses


@pytest.mark.parametrize("mode", ["deterministic", "atomic"])
def test_two_identical_steps_give_the_same_gradients(dev, mode):
    """With torch's own layers told to be deterministic (`torch.backends.cudnn.deterministic`: MIOpen's deterministic
    convolution-gradient algorithms) the deterministic scatters make a whole training step reproducible: all 104
    gradient tensors bit-identical run to run (measured with the atomic scatters on the same step: 39 of 104 tensors
    differ in their last bits -- not asserted, an atomic order may repeat).  The atomic scheme must agree to rounding."""
    from s4g_release_amd import functions as F
    F.set_backward_mode(mode)
    was = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    try:
        grads = []
        for _ in range(2):
            net, pts, target = _setup(dev, seed=3)
            _loss(net, pts, target).backward()
            grads.append({n: p.grad.clone() for n, p in net.named_parameters()})
        gmax = max(float(g.abs().max()) for g in grads[0].values())
        assert gmax > 0
        for n in grads[0]:
            a, b = grads[0][n], grads[1][n]
            if mode == "deterministic":
                assert torch.equal(a, b), n
            assert float((a - b).abs().max()) < 1e-5 * gmax, n
    finally:
        F.set_backward_mode("deterministic")
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = was
