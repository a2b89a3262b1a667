#!/usr/bin/env python3
"""Crash-safety of the whole forward on pathological clouds (round 6, after the non-finite finding): every case in a child
process of its own (a GPU memory fault aborts the process that caused it), fused path and reference-shaped modules, shipped
configuration.  Values are not judged here (the operators are checked against the oracle on such clouds by tools/fuzz_ops.py) --
only: does the call complete, are the clean scenes of the batch unaffected.   python tools/pathological_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["all_identical", "two_points", "collinear", "coplanar_lattice", "huge_coordinates", "overflowing_coordinates",
         "tiny_extent", "denormals", "one_far_outlier", "negative_zero", "n_25599", "n_30000", "n_65535", "inf_only", "nan_everywhere"]

CHILD = r'''
import sys, torch, numpy as np
sys.path.insert(0, %(root)r)
from s4g_release_amd import synth
from s4g_release_amd.fused import FusedPointNet2
from tests import golden_util as GU
case, path = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
N = {"n_25599": 25599, "n_30000": 30000, "n_65535": 65535}.get(case, 25600)
pts = synth.make_batch([0, 1, 2], N)
rng = np.random.default_rng(1)
bad = pts[1]
if case == "all_identical": bad[:] = bad[:, :1]
elif case == "two_points": bad[:, ::2] = bad[:, :1]; bad[:, 1::2] = bad[:, 1:2]
elif case == "collinear": bad[1:] = 0.0
elif case == "coplanar_lattice": bad[:] = (rng.integers(0, 30, size=(3, N)) * np.float32(0.01)).astype(np.float32); bad[2] = -1.0
elif case == "huge_coordinates": bad *= np.float32(1e15)
elif case == "overflowing_coordinates": bad *= np.float32(1e25)
elif case == "tiny_extent": bad *= np.float32(1e-20)
elif case == "denormals": bad *= np.float32(1e-42)
elif case == "one_far_outlier": bad[:, 5] = 1e6
elif case == "negative_zero": bad[:, ::3] = -0.0
elif case == "inf_only": bad[:] = np.inf
elif case == "nan_everywhere": bad[:] = np.nan
x = torch.from_numpy(pts).to(dev)
net = GU.shipped_net(dev)
with torch.no_grad():
    if path == "fused":
        run = FusedPointNet2(net)
        out = run({"scene_points": x})
        alone = run({"scene_points": x[2:3].contiguous()})
    else:
        out = net({"scene_points": x})
        alone = net({"scene_points": x[2:3].contiguous()})
torch.cuda.synchronize()
clean = all(torch.isfinite(out[k][[0, 2]]).all().item() for k in out)
same = max(float((out[k][2] - alone[k][0]).abs().max()) for k in out)
fin = all(torch.isfinite(out[k][1]).all().item() for k in out)
print("ok clean_scenes_finite=%%s scene2_vs_alone=%%.1e bad_scene_finite=%%s" %% (clean, same, fin))
'''

if __name__ == "__main__":
    for c in (sys.argv[1:] or CASES):
        for path in ("fused", "modules"):
            p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, c, path], capture_output=True, text=True, timeout=600)
            tail = (p.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in p.stderr.splitlines() if "fault" in l.lower() or "Error" in l or "Abort" in l][-2:]
            print("%-24s %-8s rc=%4d  %s  %s" % (c, path, p.returncode, tail, " | ".join(err)[:200]), flush=True)
