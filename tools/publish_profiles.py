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


def publish_extras():
    """Round 6 on: the extra tables tools/profile_r6.sh leaves (power / clock for both networks, the board's MFMA ceiling, the
    modules path's last pass, the randomized network's kernel stats, the detect loop's kernel stats)."""
    def w(name, title, body):
        with open(os.path.join(ROOT, "profiles", "r%02d_%s" % (RND, name)), "w") as f:
            f.write(title + "\n\n" + body)

    def rd(name):
        p = os.path.join(O, name)
        return open(p).read() if os.path.exists(p) else "(not captured)\n"
    w("mfma_ceiling.md", "# Round %d -- the attainable MFMA rate of the board this capture ran on (`tools/micro/mfma_ceiling 1.5`)\n\n"
      "`bench.py` keeps round 5's 1 248 (f16x2 product loop) / 1 357 (bf16 chains' loop) TFLOP/s as `roofline.attainable`." % RND, rd("mfma_ceiling.md"))
    w("power_clock.md", "# Round %d -- board power and shader clock while each contraction launch runs back to back (`tools/power_probe.py --seconds 1.5`), ONE board\n\n"
      "The same launches on the CALIBRATED network (bench.py's default since round 6: activations that carry signal) and on rounds 1-5's `randomized`\n"
      "network (every activation a per-channel constant): same power, the data-carrying operands cost CLOCK -- most visibly in the heads launch." % RND,
      "## default configuration (16 x 25 600, f16x2), `--weights calibrated`\n\n" + rd("power_clock_default.md") +
      "\n## the same, `--weights randomized` (rounds 1-5's network)\n\n" + rd("power_clock_randomized.md") +
      "\n## configs[4] (32 x 51 200, bf16), calibrated\n\n" + rd("power_clock_cfg4.md"))
    w("modules_path_kernel_stats.md", "# Round %d -- the reference-shaped modules on the HIP operators (INTEGRATION.md levels 1-2): the LAST forward pass of a kernel trace\n\n"
      "`rocprofv3 --kernel-trace --stats -- python3 bench.py --impl modules --steps 5 --warmup 2 --no-extras --no-cpu-baseline`, `tools/rocpd_last_pass.py`." % RND,
      rd("modules_last_pass.md"))
    prof = line(os.path.join(O, "bench_randomized_profiled.json"))
    if prof:
        w("randomized_kernel_stats.md", "# Round %d -- rocprofv3 --kernel-trace --stats, default configuration on rounds 1-5's `randomized` network\n\n"
          "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --weights randomized --steps 10 --warmup 2 --no-cpu-baseline --no-extras`; "
          "bench line of the profiled run: %.1f scenes/s, %.2f ms/step, contractions %.2f ms/step.  Compare `r%02d_default_kernel_stats.md` "
          "(calibrated network, same build, same box)." % (RND, prof["value"], prof["ms_per_step"], prof["roofline"]["ms_per_step"], RND),
          rd("randomized_kernel_stats.md"))
    w("detect_kernel_stats.md", "# Round %d -- rocprofv3 --kernel-trace --stats of 25 pipelined `GraspDetector` steps (`tools/detect_loop.py`: 16 raw 48 902-point clouds\n"
      "per step -> 5 selected grasps per scene)\n\n`rocprofv3 --kernel-trace --stats -- python3 tools/detect_loop.py 25`; the loop's own output: " % RND + rd("detect_loop.txt").strip(),
      rd("detect_kernel_stats.md"))


if __name__ == "__main__" and RND >= 6:
    publish_extras()
