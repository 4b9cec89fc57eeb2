#!/usr/bin/env python3
"""Where a command's time goes inside the resident evaluation kernel (abd_resident.hpp), per workgroup: needs the
diagnostic build (-DABD_STAMPS, see tools/README.md).  usage: ABD_HIP_LIB=build/libabd_hip_stamps.so
probe_resident_phases.py [chains] [iterations]"""
import ctypes
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ptr_file = tempfile.mktemp()
os.environ["ABD_STAMPS_PTR_OUT"] = ptr_file
os.environ.setdefault("ABD_PROBE_TARGET_ACCEPT", "0.995")
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, G = 10000, 200
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
smp = ctx.sampler(np.arange(C), th0, tune=10 ** 6, seed=3, gibbs=False, target_accept=float(os.environ["ABD_PROBE_TARGET_ACCEPT"]))
smp.run(10)
addr = int(open(ptr_file).read())
buf = (ctypes.c_ulonglong * (4096 * 16)).from_address(addr)
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)
st[:] = 0
_, stats = smp.run(iters)
smp.close()
nb = int((st[:2048, 6] > 0).sum())
s = st[:nb].astype(np.float64)
rounds = s[:, 6]
names = ["wait for the command (host turnaround + PCIe + relay)", "constants + power tables", "pieces: start state + gap walk",
         "reduction, partial row out, count in", "fixed-order sum + row to the host (last workgroup only), closing barrier"]
print(f"{C} chains resident, {nb} workgroups reported, {rounds.mean():.0f} commands each (summed over all kernels of the run); us per command:")
for k, nm in enumerate(names):
    v = s[:, k] / rounds / 100.0
    print(f"  {nm:75s} median {np.median(v):6.2f}  max {v.max():6.2f}  workgroup 0 {v[0]:6.2f}")
tot = s[:, :5].sum(axis=1) / rounds / 100.0
print(f"  {'sum':75s} median {np.median(tot):6.2f}")
print(f"  last-in counts: max share of one workgroup {s[:, 5].max() / rounds.mean():.2f}")

# when the kernels ran (workgroup 0's clock): how many were on the device at the same time
log = st[2048:].reshape(-1, 4)[:2048].astype(np.float64)
log = log[log[:, 1] > 0]
if len(log):
    t0 = log[:, 0].min()
    ev = sorted([(a, 1) for a in log[:, 0]] + [(b, -1) for b in log[:, 1]])
    busy = {k: 0.0 for k in range(C + 1)}
    cur, prev = 0, ev[0][0]
    for t, d in ev:
        busy[min(cur, C)] += t - prev
        cur += d
        prev = t
    span = (log[:, 1].max() - t0) / 100.0
    print(f"{len(log)} kernel launches logged over {span / 1e3:.2f} ms; share of that time with k kernels on the device: " +
          ", ".join(f"{k}: {busy[k] / 100.0 / span:.2f}" for k in range(C + 1)))
    dur = (log[:, 1] - log[:, 0]) / 100.0
    print(f"kernel lifetime: median {np.median(dur):.0f} us, commands per kernel median {np.median(log[:, 3]):.0f}; "
          f"lifetime per command median {np.median(dur / np.maximum(log[:, 3], 1)):.1f} us")
