#!/usr/bin/env python3
"""rocprofv3 --hip-trace --kernel-trace CSVs of a sampler run: how long after its hipLaunchKernel call a kernel starts on the
device, and how long after a sum kernel ends the next evaluation launch is called.  usage: trace_launch_latency.py DIR"""
import csv
import glob
import sys

import numpy as np

d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
ht = list(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
api = {r["Correlation_Id"]: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in ht if "Launch" in r["Function"]}
lat_ev, lat_fin, api_ev, api_fin = [], [], [], []
byq = {}
for r in kt:
    c = r["Correlation_Id"]
    if c not in api:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a0, a1, _ = api[c]
    name = r["Kernel_Name"]
    if "dense_kernel" in name or "obs_kernel" in name:
        lat_ev.append(s - a1)
        api_ev.append(a1 - a0)
        byq.setdefault(r["Queue_Id"], []).append((s, e, a0, a1, "ev"))
    elif "finalize" in name:
        lat_fin.append(s - a1)
        api_fin.append(a1 - a0)
        byq.setdefault(r["Queue_Id"], []).append((s, e, a0, a1, "fin"))
f = lambda v: f"median {np.median(v) / 1e3:6.2f} us, 90 % {np.percentile(v, 90) / 1e3:6.2f} us"
print(f"{len(lat_ev)} evaluation launches: hipLaunchKernel call {f(api_ev)}; call returned -> kernel starts {f(lat_ev)}")
print(f"{len(lat_fin)} sum launches:        hipLaunchKernel call {f(api_fin)}; call returned -> kernel starts {f(lat_fin)}")
turn = []
for q, ks in byq.items():
    ks.sort()
    for (s0, e0, a0, a1, k0), (s1, e1, b0, b1, k1) in zip(ks[:-1], ks[1:]):
        if k0 == "fin" and k1 == "ev":
            turn.append(b0 - e0)  # sum ended on the device -> the host calls the next evaluation launch
print(f"sum kernel ends -> host calls the next evaluation's launch (tag over PCIe + poll + NUTS): {f(turn)}")
