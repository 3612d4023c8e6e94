#!/usr/bin/env python
"""Aggregate one rocprofv3 --pmc counter per (kernel, grid size): python tools/pmc_by_kernel.py counter_collection.csv FETCH_SIZE"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2] or "anonymous namespace" not in r["Kernel_Name"] or "at::" in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    key = (name, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
    agg[key][0] += 1
    agg[key][1] += float(r["Counter_Value"])
tot = sum(v[1] for v in agg.values())
print(f"total {sys.argv[2]} = {tot * 1024 / 1e9:.2f} GB (raw KiB units x 1024)")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:46s} grid {k[1]:>9s} wg {k[2]:>4s} n={n:4d} total {v * 1024 / 1e9:7.3f} GB  per launch {v * 1024 / n / 1e6:8.2f} MB")
