#!/usr/bin/env python3
"""Sparse-list kernels at several list densities: lane per observation vs wave per individual (ABD_OBS_LANES=0).

usage: bench_sparse.py N G fill [reps_per_cell]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd._native import Context
from abdpymc_amd import synthetic

N, G, fill = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(0)
vacs = (rng.random((N, G)) < 0.7 / G).astype(np.int8)
pcr = (rng.random((N, G)) < 0.5 / G).astype(np.int8)

def obs():
    cell = np.flatnonzero(rng.random(N * G) < fill)
    cell = np.repeat(cell, reps)
    j, g = (cell // G).astype(np.int32), (cell % G).astype(np.int32)
    return g, j, rng.choice([0.0, 2.0, 4.0], size=cell.size), rng.uniform(0, 2, size=cell.size)

s_obs, n_obs = obs(), obs()
for chains in (1, 4):
    ctx = Context(G, N, s_obs, n_obs, vacs, pcr, n_chains=chains)
    for c in range(chains):
        ctx.set_discrete(c, (rng.random((G, N)) < 1 / G).astype(np.int8), (rng.random(N) < 0.5).astype(np.int8))
    th = synthetic.theta_init(G) + 0.2 * rng.standard_normal((300, chains, 17))
    ids = np.arange(chains)
    for k in range(30):
        ctx.logp_dlogp_batch(ids, th[k])
    ctx.kernel_timing(True)
    ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for k in range(30, 300):
        ctx.logp_dlogp_batch(ids, th[k])
    dt = (time.perf_counter() - t0) / 270
    ms, n = ctx.kernel_time()
    print(f"N={N} G={G} fill={fill} reps={reps} obs={s_obs[0].size + n_obs[0].size} chains={chains} "
          f"lanes={os.environ.get('ABD_OBS_LANES', '1')}: call {dt * 1e6:7.1f} us, kernel {ms / n * 1e3:7.1f} us")
    ctx.close()
