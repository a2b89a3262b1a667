#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, last dispatch."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][-70:]
        d = agg.setdefault(k, {})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d["_vgpr"] = int(r["VGPR_Count"]); d["_lds"] = int(r["LDS_Block_Size"]); d["_grid"] = int(r["Grid_Size"])
    for k, v in agg.items():
        print(k)
        print("   " + "  ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
