#!/usr/bin/env python3
"""Same (theta, discrete state) evaluated again and again with other work in between: must be the same bits."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd.data import TiterData
from abdpymc_amd.model import model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
td = TiterData.from_disk(os.path.join(ROOT, "tests", "golden", "test_cohort"))
m = model(td, n_chains=2)
ctx = m.ctx
pt = m.initial_point()
rng = np.random.default_rng(0)
th = np.stack([m.ravel(pt) + 0.1 * rng.uniform(-1, 1, 17) for _ in range(2)])
states = [((rng.random((26, 10)) < 0.08).astype(np.int8), (rng.random(10) < 0.5).astype(np.int8)) for _ in range(2)]
for c in range(2):
    ctx.set_discrete(c, *states[c])
ref = ctx.logp_dlogp_batch([0, 1], th)
bad = 0
for r in range(reps):
    mode = r % 4
    if mode == 1:
        ctx.gibbs_sweep([0, 1], th, seed=r, sweep=r)      # other kernels in between (changes the state ...)
        for c in range(2):
            ctx.set_discrete(c, *states[c])               # ... which is then restored
    elif mode == 2:
        ctx.logp_dlogp_batch([0], th[:1] + 0.01)
    elif mode == 3:
        ctx.wait()
    lp, g = ctx.logp_dlogp_batch([0, 1], th)
    if not (np.array_equal(lp, ref[0]) and np.array_equal(g, ref[1])):
        bad += 1
        if bad < 10:
            print(f"rep {r} mode {mode}: lp {lp - ref[0]}, max |dg| {np.abs(g - ref[1]).max(1)}", flush=True)
print(f"{bad} of {reps} evaluations differed")
