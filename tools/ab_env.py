#!/usr/bin/env python3
"""A/B runs of bench.py under different environment settings (GPU box):
    tools/ab_env.py VAR v1 v2 ... [-- extra bench args]
prints ms/step, the contraction total and every gemm[...] launch per setting."""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
var, vals = args[0], args[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for v in vals:
    env = dict(os.environ)
    if v != "unset":
        env[var] = v
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3",
                          "--no-extras", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(v, "FAILED", out.stderr[-1500:])
        continue
    d = json.loads(line[0])
    rows[v] = d
    print("%s=%s  ms/step %.3f  scenes/s %.1f  contractions %.3f ms  frac %.4f" % (
        var, v, d["ms_per_step"], d["value"], d["roofline"]["ms_per_step"], d["roofline"]["frac"]))
names = sorted({k for d in rows.values() for k in d["kernels"] if k.startswith(("gemm", "interp"))})
for k in names:
    print("%-60s" % k[:60] + "  ".join("%8.4f" % rows[v]["kernels"].get(k, {}).get("ms", float("nan")) for v in rows))
