#!/usr/bin/env python3
"""Copy a tools/profile_r<N>.sh capture (gpurun_out/r<N>prof, N = S4G_PROFILE_ROUND, default 3) into profiles/: kernel-stats tables with
their command header, the un-profiled bench lines; then tools/make_traffic_json.py for the PMC part."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = int(os.environ.get("S4G_PROFILE_ROUND", "3"))
O = os.path.join(ROOT, "gpurun_out", "r%dprof" % RND)
LEGEND = ("Template arguments: mlp_chain_kernel<loader, epilogue, RW, KC, PL> (loader 0 plain, 3 GATHER_MLP1, 4 GATHER_ADD, "
          "5 INTERP_ADD; epilogue 0 store, 1 max; RW 2 / 1 / 8 = 128- / 256- / 512-wide; PL 2 = f16x2, 1 = bf16), "
          "mlp_heads_kernel<PL, ring depth, 32-position blocks, FP tail in front>, mlp_gemm_f16x2_kernel<loader, epilogue, NCB, PL>, "
          "fps_pruned_kernel<threads, points per lane, fmad, index type, picks per exchange>.")


def line(path):
    rows = [l for l in open(path) if l.startswith("{")]
    return json.loads(rows[0]) if rows else None


def main():
    for tag, args, title in (("default", "--steps 10 --warmup 2 --no-cpu-baseline --no-extras", "default configuration"),
                             ("cfg4", "--points 51200 --batch 32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras",
                              "cfg4 configuration")):
        prof = line(os.path.join(O, "bench_%s_profiled.json" % tag))
        table = open(os.path.join(O, "%s_kernel_stats.md" % tag)).read()
        head = ["# Round %d -- rocprofv3 --kernel-trace --stats, %s" % (RND, title), "",
                "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py %s` (tools/profile_r%d.sh); bench line of the "
                "profiled run: %.1f scenes/s, %.2f ms/step, contractions %.2f ms/step (HIP events); un-profiled run of the "
                "same build: profiles/r%02d_bench_%s.json." % (args, RND, prof["value"], prof["ms_per_step"],
                                                               prof["roofline"]["ms_per_step"], RND, tag),
                LEGEND, ""]
        with open(os.path.join(ROOT, "profiles", "r%02d_%s_kernel_stats.md" % (RND, tag)), "w") as f:
            f.write("\n".join(head) + "\n" + table)
        raw = [l for l in open(os.path.join(O, "bench_%s.json" % tag)) if l.startswith("{")][0]
        with open(os.path.join(ROOT, "profiles", "r%02d_bench_%s.json" % (RND, tag)), "w") as f:
            f.write(raw)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py")])


if __name__ == "__main__":
    main()
