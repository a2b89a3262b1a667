"""Stand-in for `FusedPointNet2` in bench.py's TEST MODE (S4G_BENCH_BACKEND=gloo, tests/test_bench_world.py).

It computes nothing of the network -- the HIP kernels need a GPU and there is no CPU fallback of the
product path.  It only has the fast path's calling shape (`submit()` -> handle -> `result()`, `__call__`,
`precision`) and returns a `fused.PackedPred` whose 21 channels are a fill pattern of the scene's own
coordinates, so that a scene's output identifies the scene: what a test needs to see that bench.py's
all-gather put every rank's block where it belongs."""
import torch

from s4g_release_amd.dist import HEADS
from s4g_release_amd.fused import PackedPred

CHANNELS = (3, 9, 4, 5)


def outputs_of(scene_points):
    """(B, 21, N) fill pattern: channel c of scene b = x + c * y - z of that scene's points."""
    x = scene_points.float()
    c = torch.arange(sum(CHANNELS), dtype=torch.float32).view(1, -1, 1)
    return (x[:, 0:1] + c * x[:, 1:2] - x[:, 2:3]).contiguous()


class _Handle:
    def __init__(self, pred):
        self.pred = pred

    def result(self):
        return self.pred


class StubRunner:
    precision = "stub"

    def __init__(self):
        import os
        self.passes = 0
        # test hook (tests/test_bench_world.py): this rank raises at its first pass
        if os.environ.get("S4G_BENCH_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("bench_stub: rank %s told to fail" % os.environ.get("RANK"))

    def submit(self, data_batch):
        self.passes += 1
        packed = outputs_of(data_batch["scene_points"])
        return _Handle(PackedPred(zip(HEADS, packed.split(CHANNELS, dim=1)), packed=packed))

    def __call__(self, data_batch):
        return self.submit(data_batch).result()
