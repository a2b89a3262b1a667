#!/usr/bin/env python3
"""profiles/r0N_traffic.json + profiles/r0N_pmc_counters.md from the PMC csv files that
tools/profile_r<N>.sh left under gpurun_out/r<N>prof/ (N = S4G_PROFILE_ROUND, default 3) (run in the build container after the gpurun call).

HBM-side bytes per launch = FETCH_SIZE + WRITE_SIZE (KiB counters, separate passes).  gfx950's
FETCH_SIZE counts 64 B for every 128-B request of a 16-byte-per-lane streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): the contraction kernels read their
activations, W fragments and gathered rows 16 bytes per lane, so their FETCH_SIZE is DOUBLED here
("x2-corrected"); the ball-query / grouping kernels mix 4-byte plane reads, 8-byte index reads and
16-byte record gathers, so theirs is reported as measured (a lower bound).  Every entry carries the
hash of the sources it was measured on (bench.source_stamp); bench.py attaches an entry only when
that hash matches the tree it runs from."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RND = int(os.environ.get("S4G_PROFILE_ROUND", "3"))      # which round's capture (tools/profile_r<N>.sh)
O = os.path.join(ROOT, "gpurun_out", "r%dprof" % RND)
CONTRACTION = ("mlp_chain_kernel", "mlp_heads_kernel", "mlp_gemm_")


def newest(pattern):
    """gpurun merges every call's files into the same local folders: keep the latest capture only"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def per_kernel(tag, counter):
    files = newest(os.path.join(O, "pmc_%s_%s" % (tag, counter), "**", "*counter_collection.csv"))
    agg = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            n, v = agg.get(k, (0, 0.0))
            agg[k] = (n + 1, v + float(r["Counter_Value"]) * 1024.0)
    return agg


def step_totals(tag, steps):
    """bytes per step over the contraction launches; bench ran `steps` timed + 1 warm-up + extras off"""
    out = {}
    fetch, write = per_kernel(tag, "FETCH_SIZE"), per_kernel(tag, "WRITE_SIZE")
    rows = []
    tf = tw = 0.0
    for k in fetch:
        if not any(c in k for c in CONTRACTION):
            continue
        n, f = fetch[k]
        _, w = write.get(k, (n, 0.0))
        rows.append((k, n, 2.0 * f / n, w / n))
        tf += 2.0 * f
        tw += w
    launches = sum(r[1] for r in rows)
    return rows, tf / steps, tw / steps, launches / steps


def main():
    stamp = bench.source_stamp()
    entries = {}
    md = ["# Round %d -- PMC passes (rocprofv3 --pmc, one counter per pass, tools/profile_r%d.sh)" % (RND, RND), "",
          "Source stamp of the measured tree: `%s`.  FETCH_SIZE of the contraction kernels is x2-corrected "
          "(16-byte-per-lane streaming reads, see tools/make_traffic_json.py); WRITE_SIZE as measured." % stamp, ""]
    for tag, key, label in (("default", "contractions[step,B=16,N=25600,precision=f16x2]", "default bench (16 x 25 600, f16x2)"),
                            ("cfg4", "contractions[step,B=32,N=51200,precision=bf16]", "configs[4] (32 x 51 200, bf16)")):
        rows, f, w, n = step_totals(tag, 3)      # 1 warm-up + 2 timed steps
        if not rows:
            continue
        entries[key] = {"traffic_bytes": int(f + w), "fetch_bytes_x2_corrected": int(f), "write_bytes": int(w),
                        "launches_per_step": n, "source_stamp": stamp,
                        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 2 "
                                  "--warmup 1 --no-pipeline`, FETCH_SIZE x2-corrected for the 16 B/lane streams, "
                                  "tools/make_traffic_json.py, profiles/r%02d_pmc_counters.md" % RND}
        md += ["## Contraction launches of one step, %s" % label, "",
               "| kernel | launches (3 steps) | FETCH MB / launch (x2) | WRITE MB / launch |", "|---|---:|---:|---:|"]
        for k, nl, fl, wl in sorted(rows, key=lambda r: -(r[2] + r[3]) * r[1]):
            md.append("| `%s` | %d | %.1f | %.1f |" % (k[:90], nl, fl / 1e6, wl / 1e6))
        md += ["", "Per step: FETCH %.2f GB (x2-corrected) + WRITE %.2f GB = **%.2f GB**, %.0f launches." % (
            f / 1e9, w / 1e9, (f + w) / 1e9, n), ""]
    fetch, write = per_kernel("ops", "FETCH_SIZE"), per_kernel("ops", "WRITE_SIZE")
    pair = 0.0
    md += ["## ball_query + group_points (operator API, B = 16, N = 25 600, M = 5 120, K = 64; FETCH as measured)", "",
           "| kernel | launches | FETCH MB / launch | WRITE MB / launch |", "|---|---:|---:|---:|"]
    for k, (n, f) in fetch.items():
        if not any(c in k for c in ("bq_", "ball_query", "group_", "xyz_to_aos")):
            continue
        w = write.get(k, (n, 0.0))[1]
        md.append("| `%s` | %d | %.1f | %.1f |" % (k[:90], n, f / n / 1e6, w / n / 1e6))
        if "long, false, 13, 1>" in k or "bq_grid_build" in k or "group_xyz_aos" in k or "xyz_to_aos" in k:
            pair += (f + w) / n
    if pair > 0:
        entries["ball_query+group_points[N=25600,M=5120,K=64,B=16]"] = {
            "traffic_bytes": int(pair), "source_stamp": stamp,
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_ops.py --ops ball,group; "
                      "grid build + SA1-size query + xyz_to_aos + group_xyz_aos, one launch each; FETCH_SIZE as measured "
                      "(4-byte plane reads / 16-byte gathers: a lower bound), profiles/r%02d_pmc_counters.md" % RND}
        md += ["", "Pair (build + query + AoS copy + group): **%.1f MB** per launch set against 158.3 MB algorithmic." % (pair / 1e6), ""]
    # SQ counters of the dominant kernels
    for tag in ("default", "cfg4"):
        files = newest(os.path.join(O, "pmc_%s_SQ" % tag, "**", "*counter_collection.csv"))
        agg = collections.OrderedDict()
        for f in files:
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if not any(c in k for c in ("mlp_chain_kernel", "mlp_heads_kernel")):
                    continue
                d = agg.setdefault(k, collections.defaultdict(float))
                d[r["Counter_Name"]] += float(r["Counter_Value"])
                d["_n"] += 1.0 / 8
                d["_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / 8
        if not agg:
            continue
        md += ["## SQ counters, %s (`bench.py --steps 2 --warmup 1 --no-pipeline`, sums over 3 launches)" % tag, "",
               "| kernel | avg us | waves | VALU / wave (non-MFMA) | MFMA / wave | MFMA busy of SIMD time | wave time: active / issue-stalled / parked |",
               "|---|---:|---:|---:|---:|---:|---|"]
        for k, d in agg.items():
            waves = d["SQ_WAVES"]
            us = d["_us"] / max(d["_n"], 1)
            simd_cycles = 1024.0 * d["_us"] * 1e-6 * 1.9e9          # ~1.9 GHz under MFMA load (see the clock table below)
            wc = max(d["SQ_WAVE_CYCLES"], 1.0)
            md.append("| `%s` | %.0f | %.0f | %.0f | %.0f | %.0f %% | %.0f / %.0f / %.0f %% |" % (
                k[:60], us, waves / max(d["_n"], 1), (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / waves,
                d["SQ_INSTS_MFMA"] / waves, 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                100 * d["SQ_ACTIVE_INST_ANY"] / wc, 100 * d["SQ_WAIT_INST_ANY"] / wc, 100 * d["SQ_WAIT_ANY"] / wc))
        md.append("")
    # effective clock and matrix-pipe duty at that clock
    for tag in ("default", "cfg4"):
        files = newest(os.path.join(O, "pmc_%s_CLK" % tag, "**", "*counter_collection.csv"))
        agg = collections.OrderedDict()
        for f in files:
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if not any(c in k for c in ("mlp_chain_kernel", "mlp_heads_kernel", "mlp_gemm_f16x2", "fps_pruned", "fps_cluster")):
                    continue
                d = agg.setdefault(k, collections.defaultdict(float))
                d[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    d["_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    d["_n"] += 1
        if not agg:
            continue
        md += ["## Effective shader clock, %s (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / wall time; "
               "SQ_VALU_MFMA_BUSY_CYCLES / (128 SIMDs per XCD x GRBM_GUI_ACTIVE) = matrix-pipe duty at that clock)" % tag, "",
               "| kernel | launches | avg us | effective clock MHz | matrix pipe busy | share of the 2.5 PFLOP/s peak the clock leaves |",
               "|---|---:|---:|---:|---:|---:|"]
        for k, d in agg.items():
            mhz = d["GRBM_GUI_ACTIVE"] / 8.0 / max(d["_us"], 1e-9)
            md.append("| `%s` | %d | %.0f | %.0f | %.1f %% | %.2f |" % (
                k[:60], d["_n"], d["_us"] / max(d["_n"], 1), mhz,
                100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * max(d["GRBM_GUI_ACTIVE"], 1.0)), mhz / 2400.0))
        md.append("")
    with open(os.path.join(ROOT, "profiles", "r%02d_traffic.json" % RND), "w") as f:
        json.dump(entries, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "r%02d_pmc_counters.md" % RND), "w") as f:
        f.write("\n".join(md) + "\n")
    print("wrote %d traffic entries, stamp %s" % (len(entries), stamp))


if __name__ == "__main__":
    main()
