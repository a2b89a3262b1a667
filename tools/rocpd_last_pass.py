#!/usr/bin/env python3
"""Per-kernel totals of the LAST forward pass in a rocprofv3 kernel trace (rocpd database): the steady state of a
run whose first passes include library warm-up (MIOpen's solver search runs dozens of reference convolutions in the
first pass of the reference-shaped modules).  Markdown table on stdout.
Usage: rocpd_last_pass.py <db> [first-kernel substring = fps_cell_sort] [rows = 30]"""
import sqlite3
import sys


def main(path, first="fps_cell_sort", top="30"):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    i0 = starts[-1]
    acc = {}
    for name, s, e in rows[i0:]:
        n, t = acc.get(name, (0, 0.0))
        acc[name] = (n + 1, t + (e - s) / 1e3)
    total = sum(t for _, t in acc.values())
    span = (rows[-1][2] - rows[i0][1]) / 1e3
    print("last forward pass: %d launches, %.1f us of kernel time, %.1f us from its first kernel's start to its last "
          "kernel's end\n" % (len(rows) - i0, total, span))
    print("| kernel | calls | total us | avg us | % of the pass's kernel time |\n|---|---:|---:|---:|---:|")
    for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(top)]:
        short = name if len(name) <= 110 else name[:107] + "..."
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (short, n, t, t / n, 100.0 * t / total))


if __name__ == "__main__":
    main(*sys.argv[1:])
