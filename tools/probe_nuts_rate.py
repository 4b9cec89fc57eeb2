#!/usr/bin/env python3
"""logp+grad evaluations per second AS THE NATIVE SAMPLER SEES THEM (NUTS only, no sweep): leapfrogs of all chains
per wall second inside abd_sampler_run.  usage: probe_nuts_rate.py [c3|c2|default] [chains] [iterations]
ABD_SAMPLER_UNIT=B sets the chains per independent unit (default: 1 for large dense cohorts, 2-8 otherwise).
The run ends with its slowest chain: with few iterations the rate understates what a long run sees (config 3, 4 chains:
77 k evaluations/s over 40 iterations, 90 k over 1000)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 60
if cfg == "default":
    from abdpymc_amd.model import model
    from tests.test_data_loader import default_cohort

    td = default_cohort(os.path.join(ROOT, "tests", "golden"))
    m = model(td, splits=(14, 20), n_chains=C)
    ctx, G, N = m.ctx, td.n_gaps, td.n_inds
    pt = m.initial_point()
    states = [(pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))] * C
    th0 = np.stack([m.ravel(pt) + 0.1 * np.random.default_rng(c).uniform(-1, 1, 17) for c in range(C)])
else:
    N, G = {"c2": (1000, 60), "c3": (10000, 200)}[cfg]
    sc = synthetic.make_cohort(N, G)
    ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C, storage=os.environ.get("ABD_PROBE_STORAGE", "f64"))
    states = [synthetic.make_chain_state(N, G, c) for c in range(C)]
    row = int(os.environ.get("ABD_PROBE_THETA_ROW", "0"))  # 5 with 25 rows: the starting points bench.py's NUTS run uses
    th0 = np.stack([synthetic.make_thetas(G, 25 if row else 1, c)[row] for c in range(C)])
if os.environ.get("ABD_PROBE_SAME_STATE", "0") == "1":  # every chain on chain 0's discrete state (balanced trees: bench.py's `nuts`)
    states = [states[0]] * C
for c in range(C):
    ctx.set_discrete(c, *states[c])
ta = float(os.environ.get("ABD_PROBE_TARGET_ACCEPT", "0.8"))  # closer to 1: smaller steps, longer trees
gibbs = os.environ.get("ABD_PROBE_GIBBS", "0") == "1"  # 1: the compound step (sweep after every transition)
tune = int(os.environ.get("ABD_PROBE_TUNE", "100"))  # adaptation ends after this many iterations: the timed ones run at a fixed step size
smp = ctx.sampler(np.arange(C), th0, tune=tune, seed=3, gibbs=gibbs, target_accept=ta)
smp.run(max(tune, 60 if gibbs else 15))  # step size settles (and, with the sweep, the discrete state leaves its random start)
t0 = time.perf_counter()
_, st = smp.run(iters)
dt = time.perf_counter() - t0
evals = float(st["n_steps"].sum())
# chains are independent and finish at different times: the rate while ALL of them are still at work
t_first = float(st["t_done"][:, -1].min())
in_window = float(st["n_steps"][st["t_done"] <= t_first].sum())
per_chain = [int(x) for x in st["n_steps"].sum(axis=1)]
print(f"{cfg} chains={C} unit={os.environ.get('ABD_SAMPLER_UNIT', 'auto')}: {evals / dt:,.0f} evals/s as seen by NUTS over the call, "
      f"{in_window / t_first:,.0f} while all chains are at work (the first one finishes after {t_first / dt * 100:.0f} % of the call; leapfrogs per chain {per_chain}); "
      f"{evals / iters / C:.1f} leapfrogs per iteration and chain, {dt / iters * 1e3:.2f} ms of wall time per iteration of all chains, "
      f"{dt / (evals / C) * 1e6:.1f} us per leapfrog of a chain; wait fall-backs {ctx.wait_fallbacks}")
smp.close()
