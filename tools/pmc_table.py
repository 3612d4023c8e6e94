#!/usr/bin/env python
"""Per (kernel, grid) mean of every counter in one or more rocprofv3 counter_collection.csv files:
    python tools/pmc_table.py label=path.csv [label=path.csv ...]  ->  CSV rows  label,kernel,grid,counter,mean_per_dispatch,dispatches"""
import collections
import csv
import sys

print("label,kernel,grid,counter,mean_per_dispatch,dispatches")
for arg in sys.argv[1:]:
    label, path = arg.split("=", 1)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "anonymous namespace" not in k or "at::" in k:
            continue
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = (name, r.get("Grid_Size", "?"), r["Counter_Name"])
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    for (name, grid, ctr), (n, v) in sorted(agg.items()):
        print(f"{label},{name},{grid},{ctr},{v / n:.1f},{n}")
