#!/usr/bin/env python3
"""Where the timed region of bench.py goes: enqueue loop, wait, fetch (config 3)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context

N, G, C = 10000, 200, 4
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 500
th = np.stack([synthetic.make_thetas(G, K, c) for c in range(C)], axis=1)
ids = np.arange(C, dtype=np.int32)
for rep in range(4):
    ctx.wait()
    t0 = time.perf_counter()
    for k in range(K):
        ctx.enqueue(k, ids, th[k])
    t1 = time.perf_counter()
    ctx.wait()
    t2 = time.perf_counter()
    lp, g = ctx.fetch_many(np.arange(K), C)
    t3 = time.perf_counter()
    print(f"K={K}: enqueue loop {1e3 * (t1 - t0):.2f} ms, wait {1e3 * (t2 - t1):.2f} ms, fetch {1e3 * (t3 - t2):.2f} ms; total per step {1e6 * (t3 - t0) / K:.2f} us")
