#!/usr/bin/env python3
"""Where the time of one timed region of bench.py goes at small K (the driver runs K = 20): enqueue loop, abd_wait,
fetch.  usage: probe_region.py [K]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, G, C = 10000, 200, 4
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
thetas = np.stack([np.stack([synthetic.make_thetas(G, 1, 100 * k + c)[0] for c in range(C)]) for k in range(K)])
chains = np.arange(C, dtype=np.int32)


def region():
    t0 = time.perf_counter()
    for k in range(K):
        ctx.enqueue(k, chains, thetas[k])
    t1 = time.perf_counter()
    ctx.wait()
    t2 = time.perf_counter()
    lp, g = ctx.fetch_many(np.arange(K), C)
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, t3 - t0


t_w = time.perf_counter()
while time.perf_counter() - t_w < 0.1:
    region()
r = np.array([region() for _ in range(400)]) * 1e6
med = np.median(r, axis=0)
print(f"K = {K}, {ctx.n_pipes} pipes: enqueue loop {med[0]:.1f} us, wait {med[1]:.1f} us, fetch {med[2]:.1f} us, region {med[3]:.1f} us "
      f"= {med[3] / K:.2f} us per step = {K * C / med[3] * 1e6:,.0f} evals/s")
