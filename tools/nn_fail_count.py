"""Debug: how many queries of the FP3-size 3-NN grid search go to the fallback scan."""
import ctypes
import torch
from s4g_release_amd import _cabi, synth, functions as F

B, N1, N2 = 16, 25600, 5120
xyz = torch.from_numpy(synth.make_batch(list(range(B)), N1)).cuda()
idx = F.furthest_point_sample(xyz, N2) if hasattr(F, "furthest_point_sample") else None
from s4g_release_amd.fused import FusedPointNet2  # noqa
lib = _cabi.lib()
fi = torch.empty((B, N2), dtype=torch.int32, device="cuda")
ctr = torch.empty((B, 3, N2), dtype=torch.float32, device="cuda")
ws, nb = F._workspace(_cabi.S4G_OP_FPS, xyz.device, B, N1, N2, 0)
_cabi.check(lib.s4g_fps_gather_i32(xyz.data_ptr(), B, N1, N2, fi.data_ptr(), ctr.data_ptr(), F._ptr(ws), nb, F._DIST_FLAGS, F._stream()), "fps")
nbytes = lib.s4g_three_nn_grid_workspace_bytes(B, N1, N2)
w8 = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
oi = torch.empty((B, N1, 3), dtype=torch.int32, device="cuda")
ow = torch.empty((B, N1, 3), dtype=torch.float32, device="cuda")
_cabi.check(lib.s4g_three_nn_weights_grid_i32(xyz.data_ptr(), ctr.data_ptr(), B, N1, N2, 1e-8, 0.02, oi.data_ptr(), ow.data_ptr(), w8.data_ptr(), nbytes, F._DIST_FLAGS, F._stream()), "nn")
torch.cuda.synchronize()
# grid_ws_bytes(B, N2): sorted 8*N2*16 + starts 8*4100*4 + 64 + N2*16 per scene
off = B * (8 * N2 * 16 + 8 * 4100 * 4 + 64 + N2 * 16)
print("fail count:", w8[off:off + 4].view(torch.int32).item(), "of", B * N1)
