#!/usr/bin/env python3
"""Phase timeline inside ONE leapfrog-train launch of a chain that has the chip to itself (DIAGNOSTIC build:
python tools/build_variant.py stamps; ABD_HIP_LIB=$PWD/build/libabd_hip_stamps.so python tools/probe_train_stamps.py).
The stamps of the last launch of a NUTS-only run survive; printed: when each phase ends (us after the earliest workgroup's
entry), median over workgroups and the last one, the tail of the workgroup that counted in last, and the leapfrog period of
the run (the rest of it is the boundary between two launches)."""
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

N, G = 10000, 200
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=1)
ctx.set_discrete(0, *synthetic.make_chain_state(N, G, 0))
th0 = synthetic.make_thetas(G, 25, 0)[5][None, :]
smp = ctx.sampler([0], th0, tune=40, seed=3, gibbs=False)
smp.run(40)
names = {0: "entry", 1: "range + args", 2: "power tables filled", 3: "exp2 table copied", 4: "barrier passed", 6: "start state rebuilt",
         7: "gap loop done", 8: "reduction + partial store"}
rows = []
addr = int(open(tmp).read().strip())
buf = np.frombuffer((C.c_ulonglong * (4096 * 16)).from_address(addr), dtype=np.uint64)
for rep in range(8):
    buf[:] = 0
    t0 = time.perf_counter()
    _, st = smp.run(10)
    dt = time.perf_counter() - t0
    arr = buf.astype(np.float64).reshape(4096, 16)
    # the workgroups of the LAST full launch: those that entered within 10 us of the latest entry among workgroups that walked
    walked = arr[:, 8] > 0
    latest = arr[walked, 0].max()
    used = walked & (arr[:, 0] > latest - 1000.0)
    a = arr[used]
    base = a[:, 0].min()
    rows.append(((a - base) / 100.0, dt / st["n_steps"].sum() * 1e6, arr[used], base))
per = np.median([r[1] for r in rows])
a = rows[-1][0]
print(f"one chain alone, {N} x {G}: {a.shape[0]} workgroups stamped; leapfrog period of the run {per:.2f} us")
for k, nme in names.items():
    print(f"  {nme:32s} median over workgroups {np.median(a[:, k]):7.2f} us   last workgroup {a[:, k].max():7.2f} us")
raw, b = rows[-1][2], rows[-1][3]
tail = raw[raw[:, 10] > b]
if tail.size:
    t = (tail[0] - b) / 100.0
    print(f"  the workgroup that counted in last: its partial store {t[8]:.2f}, counted in + barrier {t[10]:.2f}, rows summed {t[11]:.2f}, state machine done {t[12]:.2f} us")
smp.close()
