#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a sampler run: per queue, how long the evaluation kernel and its sum take and how
long the queue is empty between them (sum end -> next evaluation start = host turnaround + launch + dispatch).
usage: trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
byq = defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", r.get("Queue_ID", "?"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in sorted(byq.items()):
    ks.sort()
    ev_d, fin_d, gap_ef, gap_fe = [], [], [], []
    for (s0, e0, n0), (s1, e1, n1) in zip(ks[:-1], ks[1:]):
        is_ev0 = "dense_kernel" in n0 or "obs_kernel" in n0
        is_fin0 = "finalize" in n0
        is_ev1 = "dense_kernel" in n1 or "obs_kernel" in n1
        is_fin1 = "finalize" in n1
        if is_ev0:
            ev_d.append(e0 - s0)
            if is_fin1:
                gap_ef.append(s1 - e0)
        if is_fin0:
            fin_d.append(e0 - s0)
            if is_ev1:
                gap_fe.append(s1 - e0)
    if len(ev_d) < 20:
        continue
    f = lambda v: f"{np.median(v) / 1e3:6.2f}" if len(v) else "   n/a"
    print(f"queue {q}: {len(ev_d)} evaluations; us (medians): evaluation kernel {f(ev_d)}, gap to its sum {f(gap_ef)}, sum {f(fin_d)}, "
          f"sum end -> next evaluation start {f(gap_fe)}; cycle {f([a + b + c + d for a, b, c, d in zip(ev_d, gap_ef, fin_d, gap_fe)])}")
