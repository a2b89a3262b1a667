#!/usr/bin/env python3
"""Kernel-by-kernel timeline of the LAST forward pass of a `bench.py --no-pipeline` kernel trace:
start (ms since the pass's first kernel), duration, queue, name.  Usage: rocpd_one_batch.py <db> <first kernel substring>"""
import sqlite3
import sys


def main(path, first="fps_cell_sort"):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, start, end, queue_id from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    i0 = starts[-1]
    t0 = rows[i0][1]
    for name, s, e, q in rows[i0:]:
        print("%8.3f %8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, name[:90]))


if __name__ == "__main__":
    main(*sys.argv[1:])
