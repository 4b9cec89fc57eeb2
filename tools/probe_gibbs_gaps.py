#!/usr/bin/env python3
"""Sweep time per proposal against the number of gaps (10 000 individuals, 4 chains, converged state): cohorts beyond 256 gaps
take the 8-word instantiation of the lane-per-proposal kernel.  usage: probe_gibbs_gaps.py [G ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context

N, C = 10000, 4
for G in [int(x) for x in sys.argv[1:]] or [200, 256, 300, 512]:
    sc = synthetic.make_cohort(N, G)
    ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
    for c in range(C):
        ctx.set_discrete(c, sc.i_true, np.ones(N, dtype=np.int8))
    th = np.tile(synthetic.truth_theta(G), (C, 1))
    ctx.gibbs_sweep(np.arange(C), th, seed=1, sweep=0)
    ts, props = [], 0
    for s in range(10):
        t0 = time.perf_counter()
        _, prop = ctx.gibbs_sweep(np.arange(C), th, seed=1, sweep=1 + s)
        ts.append(time.perf_counter() - t0)
        props = int(np.sum(prop))
    t = float(np.median(ts))
    print(f"G={G}: sweep of {C} chains {1e3 * t:.2f} ms, {props} proposals, {1e9 * t / props:.2f} ns per proposal"
          + (" (wave-per-proposal kernel)" if os.environ.get("ABD_GIBBS_V1") == "1" else ""))
    ctx.close()
