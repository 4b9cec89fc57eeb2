#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean per dispatch per (kernel, grid size)."""
import collections
import csv
import glob
import sys

for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
        agg[(r["Kernel_Name"][:70], grid, wg)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, grid, wg), v in sorted(agg.items()):
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        try:
            wgs = int(grid) // int(wg)
        except Exception:
            wgs = "?"
        print(f"{k}, grid {grid} threads = {wgs} workgroups of {wg}")
        for c, vals in sorted(v.items()):
            print(f"   {c:28s} n={len(vals):3d} mean={sum(vals) / len(vals):.5g}")
