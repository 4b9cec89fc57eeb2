#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean per dispatch per kernel."""
import collections
import csv
import glob
import sys

for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        print(k)
        for c, vals in sorted(v.items()):
            print(f"   {c:28s} n={len(vals):3d} mean={sum(vals) / len(vals):.5g}")
