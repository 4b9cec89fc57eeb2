#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: per hardware queue, the evaluation kernels' durations, the gaps between consecutive
kernels on the queue (end -> next start) and the period (start -> next start); plus how many evaluation kernels are on the
device at the same time.  usage: trace_queues.py <kernel_trace.csv> [name substring, default dense_kernel]"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "dense_kernel"
byq = defaultdict(list)
allk = []
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    byq[r.get("Queue_Id", r.get("Queue_ID", "?"))].append((s, e))
    allk.append((s, e))
pc = lambda v, q: np.percentile(v, q) / 1e3
for q, ks in sorted(byq.items()):
    ks.sort()
    if len(ks) < 20:
        continue
    ks = ks[len(ks) // 4:]  # steady state
    d = [e - s for s, e in ks]
    gap = [b[0] - a[1] for a, b in zip(ks[:-1], ks[1:])]
    per = [b[0] - a[0] for a, b in zip(ks[:-1], ks[1:])]
    print(f"queue {q}: {len(ks)} kernels; duration us p10/p50/p90 {pc(d, 10):.1f}/{pc(d, 50):.1f}/{pc(d, 90):.1f}; "
          f"gap end->next start {pc(gap, 10):.1f}/{pc(gap, 50):.1f}/{pc(gap, 90):.1f}; period {pc(per, 10):.1f}/{pc(per, 50):.1f}/{pc(per, 90):.1f}")
if allk:
    ev = sorted([(s, 1) for s, e in allk] + [(e, -1) for s, e in allk])
    t_prev, n, acc = ev[0][0], 0, defaultdict(int)
    for t, dlt in ev:
        acc[n] += t - t_prev
        t_prev, n = t, n + dlt
    tot = sum(acc.values())
    print("kernels on the device at once: " + ", ".join(f"{k}: {100.0 * v / tot:.0f} %" for k, v in sorted(acc.items())))
