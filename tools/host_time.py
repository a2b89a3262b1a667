import time, torch, sys
sys.path.insert(0, "/root/repo")
from s4g_release_amd import synth, functions as F
from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
from s4g_release_amd.fused import FusedPointNet2
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = randomize_bn_(build_pointnet2_cls(S4GConfig()), 1).to(dev).eval()
r = FusedPointNet2(net)
pts = torch.from_numpy(synth.make_batch(list(range(16)), 25600)).to(dev)
b = {"scene_points": pts}
with torch.no_grad():
    for _ in range(2): r(b)
    torch.cuda.synchronize()
    for timer in (False, True):
        F.OpTimer.reset(enabled=timer)
        t0 = time.perf_counter()
        hs = [r.submit(b) for _ in range(8)]
        t1 = time.perf_counter()
        for h in hs: h.result()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("timer=%s host submit %.2f ms/step, total %.2f ms/step" % (timer, (t1 - t0) / 8 * 1e3, (t2 - t0) / 8 * 1e3))
