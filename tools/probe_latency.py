#!/usr/bin/env python3
"""Synchronous-call latency floor: a cohort so small that the kernels are empty shells."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context
from tests.helpers import random_sparse_cohort

for name, coh in (("dense 64x4", None), ("sparse 64x4", random_sparse_cohort(64, 4, 60, 60, seed=1))):
    if coh is None:
        sc = synthetic.make_cohort(64, 4)
        ctx = Context(4, 64, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=4)
    else:
        ctx = Context(4, 64, (coh.s.idx_gap, coh.s.idx_ind, coh.s.log_dilution, coh.s.od),
                      (coh.n.idx_gap, coh.n.idx_ind, coh.n.log_dilution, coh.n.od), coh.vacs, coh.pcrpos, n_chains=4)
    rng = np.random.default_rng(0)
    for c in range(4):
        ctx.set_discrete(c, (rng.random((4, 64)) < 0.2).astype(np.int8), (rng.random(64) < 0.5).astype(np.int8))
    th = synthetic.theta_init(4) + 0.1 * rng.standard_normal((2000, 4, 17))
    ids = np.arange(4)
    for n in (1, 4):
        for k in range(200):
            ctx.logp_dlogp_batch(ids[:n], th[k, :n])
        t0 = time.perf_counter()
        for k in range(200, 2000):
            ctx.logp_dlogp_batch(ids[:n], th[k, :n])
        dt = (time.perf_counter() - t0) / 1800
        ctx.kernel_timing(True); ctx.kernel_time(reset=True)
        for k in range(200):
            ctx.logp_dlogp_batch(ids[:n], th[k, :n])
        ms, cnt = ctx.kernel_time()
        ctx.kernel_timing(False)
        print(f"{name}, {n} chain(s): {dt * 1e6:.1f} us per synchronous call; main kernel {ms / cnt * 1e3:.1f} us")
    ctx.close()
