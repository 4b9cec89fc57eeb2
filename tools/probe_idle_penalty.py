#!/usr/bin/env python3
"""How much does an idle gap before a burst of launches cost?  (config 3, 500 steps after a sleep of s seconds)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context

N, G, C, K = 10000, 200, 4, 500
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th = np.stack([synthetic.make_thetas(G, K, c) for c in range(C)], axis=1)
ids = np.arange(C, dtype=np.int32)

def burst():
    t0 = time.perf_counter()
    for k in range(K):
        ctx.enqueue(k, ids, th[k])
    ctx.wait()
    ctx.fetch_many(np.arange(K), C)
    return (time.perf_counter() - t0) / K * 1e6

for _ in range(4):
    burst()
for gap in (0.0, 0.0001, 0.0003, 0.001, 0.003, 0.01, 0.1, 0.0):
    res = []
    for rep in range(3):
        burst()
        time.sleep(gap)
        res.append(burst())
    print(f"idle {gap * 1e3:7.2f} ms before the burst: {np.round(res, 2)} us per step")
