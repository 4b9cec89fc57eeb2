#!/usr/bin/env python3
"""Gibbs sweeps on the reference's default cohort (observation lists: the wave-per-proposal kernel), for timing and
`rocprofv3 --pmc` passes.  usage: probe_gibbs_default_cohort.py [reps] [chains]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd.model import model
from tests.test_data_loader import default_cohort

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
td = default_cohort(os.path.join(ROOT, "tests", "golden"))
m = model(td, splits=(14, 20), n_chains=C)
pt = m.initial_point()
for c in range(C):
    m.ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
th = np.stack([m.ravel(pt) + 0.1 * np.random.default_rng(c).uniform(-1, 1, 17) for c in range(C)])
ids = np.arange(C)
for s in range(5):
    m.ctx.gibbs_sweep(ids, th, seed=1, sweep=s)
t0 = time.perf_counter()
for s in range(reps):
    acc, prop = m.ctx.gibbs_sweep(ids, th, seed=1, sweep=5 + s)
dt = (time.perf_counter() - t0) / reps
print(f"default cohort ({td.n_inds} individuals x {td.n_gaps} gaps), {C} chains: sweep {dt * 1e3:.3f} ms, proposals {int(np.sum(prop))}, accepted {int(np.sum(acc))}")
