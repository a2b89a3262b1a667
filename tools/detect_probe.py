import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import golden_util as GU
from s4g_release_amd import synth, postprocess as PP
from s4g_release_amd.fused import FusedPointNet2
from s4g_release_amd.detector import GraspDetector
dev=torch.device('cuda:0')
net=GU.shipped_net(dev)
run=FusedPointNet2(net)
B=16
t_raw=synth.make_batch(list(range(B)),48902)
raw=torch.from_numpy(np.ascontiguousarray(np.stack([t_raw[:,1],t_raw[:,0],-t_raw[:,2]],axis=1))).to(dev)
det=GraspDetector(run, topk=2048)
with torch.no_grad():
    probe=PP.expected_score(run({"scene_points":det.pre_processing(raw)})["score"].contiguous(),"detector")
    thr=float(torch.quantile(probe.flatten().float()[::7],0.98))
kw=dict(num_selected=5, score_threshold=thr, verticalness_threshold=-2.0)
pts=det.pre_processing(raw)
def loop(submit, n=30, warm=5, inflight=2):
    pend=[]
    def go(k):
        for _ in range(k):
            pend.append(submit())
            if len(pend)>inflight: pend.pop(0).result()
        while pend: pend.pop(0).result()
    go(warm); torch.cuda.synchronize(); t=time.perf_counter(); go(n); torch.cuda.synchronize()
    return 1e3*(time.perf_counter()-t)/n
class H:
    def __init__(s,f): s.f=f
    def result(s): return s.f()
with torch.no_grad():
    print("forward topk only          %.3f ms" % loop(lambda: run.submit({"scene_points":pts}, topk=2048)))
    print("kept + decode_top_poses    %.3f ms" % loop(lambda: (lambda h: H(lambda: PP.decode_top_poses(h.result(), pts, 50)))(run.submit({"scene_points":pts}, topk=2048))))
    print("detect full                %.3f ms" % loop(lambda: det.submit(raw, **kw)))
    print("detect no collision        %.3f ms" % loop(lambda: det.submit(raw, collision_check=False, **kw)))
    orig=det.pre_processing
    det.pre_processing=lambda c, seed=None: pts
    print("detect, pts resident       %.3f ms" % loop(lambda: det.submit(raw, **kw)))
    print("detect, resident, no coll  %.3f ms" % loop(lambda: det.submit(raw, collision_check=False, **kw)))
    det.pre_processing=orig
    real=PP.view_non_collision
    PP.view_non_collision=lambda H, c, g=None, inverse="se3", count=None: (torch.ones(H.shape[:2], dtype=torch.bool, device=H.device) & (torch.arange(H.shape[1], device=H.device).view(1,-1) < count.view(-1,1)), None)
    print("detect, collision kernel stubbed %.3f ms" % loop(lambda: det.submit(raw, **kw)))
    PP.view_non_collision=real
    import s4g_release_amd.detector as D
    realsort=torch.sort
    print("detect again               %.3f ms" % loop(lambda: det.submit(raw, **kw)))
    print("forward topk again         %.3f ms" % loop(lambda: run.submit({"scene_points":pts}, topk=2048)))
    # host time of one submit+result without waiting for the device
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): det.submit(raw, **kw).result()
    host=1e3*(time.perf_counter()-t)/10; torch.cuda.synchronize()
    print("host ms per detect call    %.3f" % host)
