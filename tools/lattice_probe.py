import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from s4g_release_amd import synth, functions as F
from s4g_release_amd.fused import FusedPointNet2
from s4g_release_amd.model import S4GConfig, build_pointnet2_cls, randomize_bn_
dev = torch.device('cuda:0')
net = build_pointnet2_cls(S4GConfig()); randomize_bn_(net, 1); net = net.to(dev).eval()
r = FusedPointNet2(net)
for variant in ('tabletop-v1', 'lattice', 'dup-heavy'):
    pts = torch.from_numpy(synth.make_batch([0, 1], 25600, variant=variant)).to(dev)
    for _ in range(2): r({"scene_points": pts})
    torch.cuda.synchronize()
    F.OpTimer.reset(enabled=True)
    t0 = time.perf_counter()
    for _ in range(3): r({"scene_points": pts})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    F.OpTimer.enabled = False
    s = F.OpTimer.summary()
    print(variant, "%.2f ms per forward" % (dt * 1e3))
    for k, v in sorted(s.items(), key=lambda kv: -kv[1][1])[:8]:
        print("   %-60s %.3f ms" % (k, v[1]))
