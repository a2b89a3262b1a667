import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from s4g_release_amd import synth, postprocess as PP
dev=torch.device('cuda:0')
B,K,N=16,2048,48902
rng=np.random.default_rng(1)
cloud=torch.from_numpy(synth.make_batch(list(range(B)),N)).to(dev)
pred={k: torch.from_numpy(rng.standard_normal((B,c,25600)).astype(np.float32)).to(dev) for k,c in (("score",3),("frame_R",9),("frame_t",4))}
pts=cloud[:,:,:25600].contiguous()
H,_,_=PP.decode_top_poses(pred, pts, K)
count=torch.full((B,),26,device=dev)
score=torch.rand(B,K,device=dev); index=torch.randint(0,25600,(B,K),device=dev)
def tm(name, f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print("%-34s %.3f ms" % (name, e0.elapsed_time(e1)/n))
tm("se3_inverse", lambda: PP.se3_inverse(H))
tm("view_non_collision (se3, count)", lambda: PP.view_non_collision(H, cloud, inverse="se3", count=count))
ok,_=PP.view_non_collision(H, cloud, inverse="se3", count=count)
tm("sort stable uint8", lambda: torch.sort((~ok).to(torch.uint8), dim=1, stable=True))
order=torch.sort((~ok).to(torch.uint8), dim=1, stable=True)[1]
tm("gather H", lambda: torch.gather(H,1,order.view(B,K,1,1).expand(-1,-1,4,4)))
tm("gather score+index", lambda: (torch.gather(score,1,order), torch.gather(index,1,order)))
live=torch.arange(K,device=dev).view(1,K)<count.view(B,1)
tm("3 x where", lambda: (torch.where(live.view(B,K,1,1),H,torch.zeros_like(H)), torch.where(live,score,torch.zeros_like(score)), torch.where(live,index,torch.full_like(index,-1))))
tm("importance_sampling", lambda: PP.importance_sampling(score, count, 5))
tm("detect_poses (full preds)", lambda: PP.detect_poses(pred, pts, 0.9, -2.0, max_poses=K))
