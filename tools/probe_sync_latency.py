#!/usr/bin/env python3
"""Latency of a synchronous logp+grad call at config 3 (10 000 x 200, fp64) for 1, 2 and 4 chains per call: what a Python-level
NUTS (PyMC through the Op) pays per leapfrog.  usage: probe_sync_latency.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

N, G, C = 10000, 200, 4
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th = np.stack([np.stack([synthetic.make_thetas(G, 1, 10 * k + c)[0] for c in range(C)]) for k in range(300)])
ids = np.arange(C)
for n in (1, 2, 4):
    for k in range(100):
        ctx.logp_dlogp_batch(ids[:n], th[k, :n])
    t0 = time.perf_counter()
    for rep in range(5):
        for k in range(300):
            ctx.logp_dlogp_batch(ids[:n], th[k, :n])
    dt = (time.perf_counter() - t0) / 1500
    print(f"{n} chain(s) per synchronous call: {dt * 1e6:.1f} us per call = {n / dt:,.0f} evals/s")
print("wait fall-backs", ctx.wait_fallbacks)
