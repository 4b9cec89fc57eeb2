#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: per kernel name total time, count, mean; and how much of the span between the first
and the last kernel the device runs 0, 1, 2, ... kernels.  usage: trace_busy.py <kernel_trace.csv> [t0_fraction t1_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
T0, T1 = ks[0][0], max(k[1] for k in ks)
lo, hi = T0 + f0 * (T1 - T0), T0 + f1 * (T1 - T0)
ks = [k for k in ks if k[0] >= lo and k[1] <= hi]
tot = defaultdict(lambda: [0, 0])
for s, e, n in ks:
    tot[n][0] += e - s
    tot[n][1] += 1
span = max(k[1] for k in ks) - ks[0][0]
print(f"span {span / 1e6:.2f} ms, {len(ks)} kernels")
for n, (t, c) in sorted(tot.items(), key=lambda x: -x[1][0]):
    print(f"  {n:60s} {t / 1e6:9.2f} ms ({100.0 * t / span:5.1f} % of the span) {c:7d} x {t / c / 1e3:8.1f} us")
ev = sorted([(s, 1) for s, e, n in ks] + [(e, -1) for s, e, n in ks])
acc, n, tp = defaultdict(int), 0, ev[0][0]
for t, d in ev:
    acc[n] += t - tp
    tp, n = t, n + d
print("kernels on the device at once: " + ", ".join(f"{k}: {100.0 * v / span:.0f} %" for k, v in sorted(acc.items())))
