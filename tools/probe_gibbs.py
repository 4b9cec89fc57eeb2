#!/usr/bin/env python3
"""Full-size Gibbs sweeps (config 3: 10 000 x 200, 4 chains) for timing / rocprofv3 counter passes.
usage: probe_gibbs.py [reps] [truth|random] [chains]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context

N, G = 10000, 200
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
if len(sys.argv) > 2 and sys.argv[2] == "truth":
    # a converged chain: the simulation's own parameters and infections
    t = synthetic.truth_theta(G)
    th = np.tile(t, (C, 1))
    for c in range(C):
        ctx.set_discrete(c, sc.i_true, np.ones(N, dtype=np.int8))
ctx.gibbs_sweep(np.arange(C), th, seed=1, sweep=0)
t0 = time.perf_counter()
for s in range(reps):
    ctx.gibbs_sweep(np.arange(C), th, seed=1, sweep=1 + s)
print(f"{C} chains: sweep {1e3 * (time.perf_counter() - t0) / reps:.2f} ms")
