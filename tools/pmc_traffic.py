#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter (FETCH_SIZE / WRITE_SIZE, KiB) over the dispatches whose
kernel name contains argv[3] (default: mlp_gemm), per name and in total.
Usage: tools/pmc_traffic.py <counter_collection.csv> <counter> [pattern] [steps]"""
import collections
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "mlp_gemm"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tot = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter or pat not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].split("(")[0][-60:]
    n, v = tot.get(k, (0, 0.0))
    tot[k] = (n + 1, v + float(r["Counter_Value"]))
g = 0.0
for k, (n, v) in tot.items():
    print("%-62s launches %4d  %10.1f MB/launch" % (k, n, v * 1024 / n / 1e6))
    g += v * 1024
print("total %.3f GB over %d step(s) -> %.3f GB/step" % (g / 1e9, steps, g / 1e9 / steps))
