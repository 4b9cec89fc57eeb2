#!/usr/bin/env python3
"""Where a dense evaluation kernel's time goes, phase by phase (DIAGNOSTIC build only: hipcc -DABD_STAMPS ... -o
build/libabd_hip_stamps.so; ABD_HIP_LIB=build/libabd_hip_stamps.so python tools/probe_stamps.py [chains] [N] [G]).
Wave 0 of every workgroup of grid row 0 records s_memrealtime (100 MHz) at phase boundaries; printed: median / max over
workgroups of each phase's end, in us after the earliest workgroup's start."""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tmp = tempfile.mktemp()
os.environ["ABD_STAMPS_PTR_OUT"] = tmp
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 200
BLOCKS = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # grid of the synchronous launch (0: the library's)
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C_)
if BLOCKS:
    ctx.set_launch_config(blocks=BLOCKS)
for c in range(C_):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th = np.stack([synthetic.make_thetas(G, 40, c) for c in range(C_)], axis=1)
chains = np.arange(C_, dtype=np.int32)
for k in range(20):
    ctx.logp_dlogp_batch(chains, th[k])
addr = int(open(tmp).read().strip())
buf = (C.c_ulonglong * (4096 * 16)).from_address(addr)
names = ["entry", "range + args", "power tables filled", "exp2 table copied", "barrier passed", "rows loaded + constrain (last piece)",
         "start state rebuilt (last piece)", "gap loop done", "reduction + partial store"]
acc = []
for k in range(20, 40):
    arr = np.frombuffer(buf, dtype=np.uint64)
    arr[:] = 0
    t0 = time.perf_counter()
    ctx.logp_dlogp_batch(chains, th[k])
    host_us = (time.perf_counter() - t0) * 1e6
    st = np.array(arr, dtype=np.float64).reshape(4096, 16)
    used = st[:, 0] > 0
    st = st[used]
    base = st[:, 0].min()
    acc.append(((st[:, :9] - base) / 100.0, host_us))
print(f"{C_} chain(s), {N} x {G}: {acc[0][0].shape[0]} workgroups; host time of the call {np.median([a[1] for a in acc]):.1f} us")
med = np.median(np.stack([np.median(a[0], axis=0) for a in acc]), axis=0)
mx = np.median(np.stack([a[0].max(axis=0) for a in acc]), axis=0)
for k, nme in enumerate(names):
    print(f"  {nme:40s} median over workgroups {med[k]:7.2f} us   last workgroup {mx[k]:7.2f} us")
# distribution of workgroup lifetimes of the last call
st = acc[-1][0]
life = st[:, 8] - st[:, 0]
print("workgroup lifetime (entry -> end) percentiles 5/25/50/75/95/100:", np.round(np.percentile(life, [5, 25, 50, 75, 95, 100]), 2))
g0s = np.array(arr, dtype=np.float64).reshape(4096, 16)[used][:, 9]
ss_t = st[:, 6] - st[:, 4]
for lo, hi in ((0, 1), (1, 50), (50, 100), (100, 150), (150, 1000)):
    sel = (g0s >= lo) & (g0s < hi)
    if sel.any():
        print(f"start gap in [{lo}, {hi}): {int(sel.sum())} workgroups (wave 0), barrier -> start state rebuilt median {np.median(ss_t[sel]):.2f} us, max {ss_t[sel].max():.2f} us; end median {np.median(st[sel, 8]):.2f} us")
print("entry time percentiles 5/25/50/75/95/100:", np.round(np.percentile(st[:, 0], [5, 25, 50, 75, 95, 100]), 2))
print("end time percentiles 5/25/50/75/95/100:", np.round(np.percentile(st[:, 8], [5, 25, 50, 75, 95, 100]), 2))
loop = st[:, 7] - st[:, 6]
print("gap loop duration (last piece) percentiles 5/50/95/100:", np.round(np.percentile(loop, [5, 50, 95, 100]), 2))
nb = st.shape[0]
for x in range(8):
    sel = np.arange(nb) % 8 == x
    print(f"  blockIdx.x % 8 == {x}: entry median {np.median(st[sel, 0]):6.2f}  end median {np.median(st[sel, 8]):6.2f}  end max {st[sel, 8].max():6.2f}")
order = np.argsort(st[:, 8])
print("slowest 8 workgroups (blockIdx.x, entry, end):", [(int(i), round(float(st[i, 0]), 1), round(float(st[i, 8]), 1)) for i in order[-8:]])
print("fastest 8 workgroups (blockIdx.x, entry, end):", [(int(i), round(float(st[i, 0]), 1), round(float(st[i, 8]), 1)) for i in order[:8]])
