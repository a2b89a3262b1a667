#!/usr/bin/env python3
"""20 pipelined `GraspDetector` steps (16 raw 48 902-point clouds each) for a kernel trace: python tools/detect_loop.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4g_release_amd import postprocess as PP, synth  # noqa: E402
from s4g_release_amd.detector import GraspDetector  # noqa: E402
from s4g_release_amd.fused import FusedPointNet2  # noqa: E402
from tests import golden_util as GU  # noqa: E402

dev = torch.device("cuda:0")
run = FusedPointNet2(GU.shipped_net(dev))
t = synth.make_batch(list(range(16)), 48902)
raw = torch.from_numpy(np.ascontiguousarray(np.stack([t[:, 1], t[:, 0], -t[:, 2]], axis=1))).to(dev)
det = GraspDetector(run, topk=2048)
with torch.no_grad():
    probe = PP.expected_score(run({"scene_points": det.pre_processing(raw)})["score"].contiguous(), "detector")
    thr = float(torch.quantile(probe.flatten().float()[::7], 0.98))
    kw = dict(num_selected=5, score_threshold=thr, verticalness_threshold=-2.0)
    pend = []
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
        pend.append(det.submit(raw, **kw))
        if len(pend) > 2:
            pend.pop(0).result()
    while pend:
        out = pend.pop(0).result()
torch.cuda.synchronize()
print("candidates per scene (after the collision check):", out.candidates[3].tolist())
