#!/usr/bin/env python3
"""Per-stream timeline of a rocprofv3 rocpd database (`--kernel-trace`): for every
stream/queue the busy time, the idle gaps between consecutive kernels and the
longest gaps, over the last `--steps` fraction of the trace.
Usage: tools/rocpd_timeline.py <results.db> [t0_frac]"""
import sqlite3
import sys


def main(path, frac=0.5):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = [t for t in tabs if t == "kernels"]
    if not view:
        print("tables:", tabs)
        return
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    q = "select name, start, end, queue_id, stream_id from kernels order by start" if "stream_id" in cols \
        else "select name, start, end, queue_id, queue_id from kernels order by start"
    rows = list(c.execute(q))
    t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
    cut = t_lo + (t_hi - t_lo) * frac
    rows = [r for r in rows if r[1] >= cut]
    print("window %.3f ms, %d kernels" % ((t_hi - cut) / 1e6, len(rows)))
    streams = {}
    for name, s, e, q, st in rows:
        streams.setdefault((q, st), []).append((s, e, name))
    for key, ks in sorted(streams.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2][:60], ks[i + 1][2][:60]) for i in range(len(ks) - 1)]
        span = ks[-1][1] - ks[0][0]
        print("queue/stream %s: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms" % (
            key, len(ks), span / 1e6, busy / 1e6, (span - busy) / 1e6))
        for g, a, b in sorted(gaps, reverse=True)[:8]:
            print("    gap %8.1f us  after %s -> %s" % (g / 1e3, a, b))
        small = [g for g, _, _ in gaps if g < 100e3]
        if small:
            print("    gaps < 100 us: n=%d mean %.1f us" % (len(small), sum(small) / len(small) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
