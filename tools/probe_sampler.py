#!/usr/bin/env python3
"""Where the native sampler's time goes on the default cohort (development probe)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd.model import model
from tests.test_data_loader import default_cohort

td = default_cohort(os.path.join(ROOT, "tests", "golden"))
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = model(td, splits=(14, 20), n_chains=chains)
pt = m.initial_point()
for gibbs in (False, True):
    q0 = np.empty((chains, 17))
    for c in range(chains):
        m.ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
        q0[c] = m.ravel(pt) + 0.1 * np.random.default_rng([0, c]).uniform(-1, 1, 17)
    smp = m.ctx.sampler(np.arange(chains), q0, tune=600, seed=1, gibbs=gibbs)
    for phase in range(8):
        t0 = time.perf_counter()
        th, st = smp.run(100)
        dt = time.perf_counter() - t0
        evals = st["n_steps"].max(0).sum()  # lock-step launches
        print(f"gibbs={gibbs} iters {phase*100:4d}-{phase*100+99:4d}: {dt:6.3f} s, depth {st['tree_depth'].mean():.2f}, "
              f"steps {st['n_steps'].mean():6.1f}, launches {int(evals)}, {dt/ (evals + 100*(1+gibbs)) * 1e6:6.1f} us/launch-ish, "
              f"eps {st['step_size'][:, -1].round(3)}, acc {st['mean_tree_accept'].mean():.2f}, div {int(st['diverging'].sum())}")
    print("inv_mass", np.round(smp.adaptation(0)[0], 4))
    smp.close()
m.close()
