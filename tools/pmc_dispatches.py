#!/usr/bin/env python3
"""Per-dispatch view of rocprofv3 --pmc counter_collection.csv: one line per dispatch of
the kernels whose name contains argv[2] (counters side by side, duration from the trace)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    d = agg.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0][-60:]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in agg.items():
    print(k, v.pop("name"), "  ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
