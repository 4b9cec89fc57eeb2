#!/usr/bin/env python3
"""Sweep time of ONE chain against the number of individuals (converged state): the fixed part of a sweep launch (tables,
the longest individual, the host's calls) and the part that grows with the cohort.  usage: probe_gibbs_scaling.py [G]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context

G = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for N in (64, 1000, 2500, 5000, 10000, 20000, 40000):
    sc = synthetic.make_cohort(N, G)
    ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=1)
    ctx.set_discrete(0, sc.i_true, np.ones(N, dtype=np.int8))
    th = synthetic.truth_theta(G)[None, :]
    ctx.gibbs_sweep([0], th, seed=1, sweep=0)
    ts = []
    for s in range(20):
        t0 = time.perf_counter()
        ctx.gibbs_sweep([0], th, seed=1, sweep=1 + s)
        ts.append(time.perf_counter() - t0)
    print(f"N={N}: sweep {1e3 * np.median(ts):.3f} ms (min {1e3 * min(ts):.3f})")
    ctx.close()
