import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from s4g_release_amd import synth, functions as F
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.make_batch(list(range(16)), 25600)).to(dev)
ctr = F.gather_points(pts, F.farthest_point_sample(pts, 5120))
idx, cnt = F.ball_query(pts, ctr, 0.02, 64)
c = cnt.cpu().numpy().ravel()
print("SA1 cnt: mean %.1f  frac==64 %.3f  pct10/50/90 %s" % (c.mean(), (c == 64).mean(), np.percentile(c, [10, 50, 90])))
# true hit counts (uncapped) via K=512 query
idx2, cnt2 = F.ball_query(pts, ctr, 0.02, 512)
c2 = cnt2.cpu().numpy().ravel()
print("uncapped hits: mean %.1f max %d  frac>64 %.3f frac>128 %.3f frac>192 %.4f" % (c2.mean(), c2.max(), (c2 > 64).mean(), (c2 > 128).mean(), (c2 > 192).mean()))
