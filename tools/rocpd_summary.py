#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite database (`--kernel-trace --stats`) into the
per-kernel summary table committed under profiles/ (name, calls, total/avg µs,
share).  Usage: tools/rocpd_summary.py <results.db> [> profiles/xxx.md]"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage "
                          "from top_kernels order by total_duration desc"))
    unit = 1.0   # durations are in microseconds in the view
    total = sum(r[2] for r in rows)
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, tot, avg, pct in rows[:top]:
        short = name if len(name) < 110 else name[:107] + "..."
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (short, calls, tot * unit, avg * unit, pct))
    print("\nkernels: %d distinct, total GPU kernel time %.1f us" % (len(rows), total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
