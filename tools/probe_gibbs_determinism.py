#!/usr/bin/env python3
"""Is a sweep from the same (state, theta, seed, sweep) always the same state?  Many random cases, each run twice
with different work before it (sparse test cohort, 2 chains)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd.data import TiterData
from abdpymc_amd.model import model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
td = TiterData.from_disk(os.path.join(ROOT, "tests", "golden", "test_cohort"))
splits = (14, 20) if os.environ.get("PROBE_SPLITS") else None
m = model(td, splits=splits, n_chains=2)
ctx = m.ctx
pt = m.initial_point()
rng = np.random.default_rng(0)
bad = 0
for r in range(reps):
    rate = rng.uniform(0, 0.3)
    st = [((rng.random((26, 10)) < rate).astype(np.int8), (rng.random(10) < 0.5).astype(np.int8)) for _ in range(2)]
    th = np.stack([m.ravel(pt) + 0.3 * rng.standard_normal(17) for _ in range(2)])
    out = []
    for rep in range(2):
        for c in range(2):
            ctx.set_discrete(c, *st[c])
        if rep == 1:
            ctx.logp_dlogp_batch([0, 1], th)
            ctx.logp_dlogp_batch([1], th[1:])
            ctx.wait()
        acc, prop = ctx.gibbs_sweep([0, 1], th, seed=r, sweep=r)
        got = [ctx.get_discrete(c) for c in range(2)]
        out.append((acc.tolist(), prop.tolist(), [g[0].tobytes() for g in got], [g[1].tobytes() for g in got]))
    if out[0] != out[1]:
        bad += 1
        what = [k for k in range(4) if out[0][k] != out[1][k]]
        if bad < 12:
            print(f"case {r} (rate {rate:.2f}): differs in {what} (0 acc, 1 prop, 2 i_raw, 3 waner): acc {out[0][0]} vs {out[1][0]}", flush=True)
print(f"{bad} of {reps} cases differed")
