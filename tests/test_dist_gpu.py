"""The multi-GPU data path with the REAL forward (one rank through RCCL; `tests/test_dist.py` covers world 2 / 3 / 4 / 8
on gloo with a stand-in forward, `tests/test_bench_world.py` covers bench.py's control flow): `dist.sharded_forward`
over `FusedPointNet2`, the packed `(B, 21, N)` payload handed to `all_gather_into_tensor` without a copy, the side-stream
`OutputGather`, and `gather_check` on device tensors."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from s4g_release_amd import dist as sdist, synth
from s4g_release_amd.fused import FusedPointNet2, PackedPred
from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
rank, world, local = sdist.init_from_env()          # WORLD_SIZE=1: no group yet
assert world == 1 and os.environ.get("NCCL_MAX_NCHANNELS") is None
os.environ.update(WORLD_SIZE="1", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r)
assert sdist.bound_rccl_channels() == "8"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
from tests import golden_util as GU
net = GU.shipped_net(dev)
run = FusedPointNet2(net)
pts = torch.from_numpy(synth.make_batch([0, 1, 2], 25600)).to(dev)
with torch.no_grad():
    direct = run({"scene_points": pts})
    assert isinstance(direct, PackedPred)
    packed, chans = sdist.pack_outputs(direct)
    assert packed.data_ptr() == direct.packed.data_ptr() and chans == [3, 9, 4, 5]      # zero-copy payload
    gathered = sdist.sharded_forward(run, pts)
    g = sdist.OutputGather("heads", device=dev)
    side = g(run({"scene_points": pts}), pts)
    rep = sdist.gather_check(g, run({"scene_points": pts}), pts)
torch.cuda.synchronize()
for k in sdist.HEADS:
    assert torch.equal(gathered[k], direct[k]) and torch.equal(side[k], direct[k]), k
assert g.last_stream.startswith("side stream") and g.payload_bytes == 3 * 21 * 25600 * 4
assert rep["blocks_verified"] == 1 and rep["own_block_bit_identical"]
dist.destroy_process_group()
print("DIST_GPU_OK")
"""


def test_sharded_forward_with_the_real_forward_over_rccl():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NCCL_MAX_NCHANNELS")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": port}], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DIST_GPU_OK" in out.stdout, out.stderr[-3000:]
