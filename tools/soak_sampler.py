"""Soak test of the native sampler on one GPU: long compound runs on the default cohort (observation lists, host threads,
leapfrog trains) and on config 3 (dense, trains, lane-per-proposal sweeps); asserts finite draws and zero wait fall-backs."""
import os, sys, time
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context
from abdpymc_amd.model import model
from tests.test_data_loader import default_cohort
td = default_cohort(os.path.join(ROOT, "tests", "golden"))
for C in (4, 6):
    m = model(td, splits=(14, 20), n_chains=C)
    pt = m.initial_point()
    for c in range(C):
        m.ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
    th0 = np.stack([m.ravel(pt) + 0.5 * np.random.default_rng(c).uniform(-1, 1, 17) for c in range(C)])
    smp = m.ctx.sampler(np.arange(C), th0, tune=2000, seed=11, gibbs=True)
    t0 = time.perf_counter(); n = 0
    for rep in range(10):
        th, st = smp.run(2000); n += 2000
        assert np.isfinite(th).all() and np.isfinite(st["lp"]).all()
    dt = time.perf_counter() - t0
    print(f"default cohort {C} chains x {n} iterations: {dt:.1f} s, {C*n/dt:.0f} chain-it/s, fallbacks {m.ctx.wait_fallbacks}, diverging {int(st['diverging'].sum())}, mean steps {st['n_steps'].mean():.1f}", flush=True)
    assert m.ctx.wait_fallbacks == 0
    smp.close(); m.close()
sc = synthetic.make_cohort(10000, 200)
for C in (4, 8):
    ctx = Context(200, 10000, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
    for c in range(C):
        ctx.set_discrete(c, *synthetic.make_chain_state(10000, 200, c))
    th0 = np.stack([synthetic.make_thetas(200, 1, c)[0] for c in range(C)])
    smp = ctx.sampler(np.arange(C), th0, tune=300, seed=5, gibbs=True)
    t0 = time.perf_counter()
    for rep in range(6):
        th, st = smp.run(100)
        assert np.isfinite(th).all() and np.isfinite(st["lp"]).all()
        print(f"  c3 {C} chains: {100*(rep+1)} iterations, {time.perf_counter()-t0:.1f} s, mean steps {st['n_steps'].mean():.0f}, fallbacks {ctx.wait_fallbacks}", flush=True)
    assert ctx.wait_fallbacks == 0
    smp.close(); ctx.close()
print("soak ok")
