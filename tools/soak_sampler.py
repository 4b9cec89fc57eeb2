#!/usr/bin/env python3
"""Long runs of the native sampler (many millions of launches): every lp finite, no polled wait ever fell back to a stream
synchronise, and a second run with the same seed gives the same bits.  usage: soak_sampler.py [default|c3] [chains] [iters]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5000


def run():
    if cfg == "default":
        from abdpymc_amd.model import model
        from tests.test_data_loader import default_cohort

        td = default_cohort(os.path.join(ROOT, "tests", "golden"))
        m = model(td, splits=(14, 20), n_chains=C)
        ctx = m.ctx
        pt = m.initial_point()
        for c in range(C):
            ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
        th0 = np.stack([m.ravel(pt) + 0.1 * np.random.default_rng(c).uniform(-1, 1, 17) for c in range(C)])
    else:
        N, G = 10000, 200
        sc = synthetic.make_cohort(N, G)
        ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
        for c in range(C):
            ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
        th0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
    smp = ctx.sampler(np.arange(C), th0, tune=iters // 2, seed=17)
    t0 = time.perf_counter()
    th, st = smp.run(iters)
    dt = time.perf_counter() - t0
    fb = ctx.wait_fallbacks
    smp.close()
    return th, st, fb, dt


th1, st1, fb1, dt1 = run()
th2, st2, fb2, dt2 = run()
evals = float(st1["n_steps"].sum())
print(f"{cfg}, {C} chains x {iters} iterations: {dt1:.1f} s and {dt2:.1f} s, {evals / dt1:,.0f} evaluations/s; lp finite: "
      f"{bool(np.all(np.isfinite(st1['lp'])))}; divergences {int(st1['diverging'].sum())}; wait fall-backs {fb1} + {fb2}; "
      f"second run bit-identical: {bool(np.array_equal(th1, th2) and np.array_equal(st1['lp'], st2['lp']))}")
assert np.all(np.isfinite(st1["lp"])) and fb1 == 0 and fb2 == 0 and np.array_equal(th1, th2)
