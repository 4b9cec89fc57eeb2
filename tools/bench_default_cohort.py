#!/usr/bin/env python3
"""logp+grad rate on the reference's default cohort (BASELINE config 1: N=1520, G=31, 35 709 observations; sparse kernel)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd.model import model  # noqa: E402
from tests.test_data_loader import default_cohort  # noqa: E402

td = default_cohort(os.path.join(ROOT, "tests", "golden"))
for chains in (1, 4):
    m = model(td, splits=(14, 20), n_chains=chains)
    rng = np.random.default_rng(0)
    q0 = m.ravel(m.initial_point())
    for c in range(chains):
        m.ctx.set_discrete(c, (rng.random((td.n_gaps, td.n_inds)) < 1 / td.n_gaps).astype(np.int8),
                           (rng.random(td.n_inds) < 0.5).astype(np.int8))
    th = q0 + 0.2 * rng.standard_normal((600, chains, 17))
    ids = np.arange(chains)
    for k in range(50):
        m.ctx.logp_dlogp_batch(ids, th[k])
    t0 = time.perf_counter()
    for k in range(50, 550):
        m.ctx.logp_dlogp_batch(ids, th[k])
    dt = time.perf_counter() - t0
    m.ctx.kernel_timing(True)
    m.ctx.kernel_time(reset=True)
    for k in range(50, 550):
        m.ctx.enqueue(k % 1000, ids, th[k])
    m.ctx.wait()
    ms, n = m.ctx.kernel_time()
    t1 = time.perf_counter()
    for s in range(20):
        m.ctx.gibbs_sweep(ids, th[s], seed=1, sweep=s)
    dg = (time.perf_counter() - t1) / 20
    print(f"default cohort, {chains} chain(s): {500 * chains / dt:9.0f} evals/s synchronous ({dt / 500 * 1e6:6.1f} us per call), "
          f"kernel {ms / n * 1e3:6.1f} us, algorithmic bytes {m.ctx.algorithmic_bytes(chains)}; Gibbs sweep {dg * 1e3:.2f} ms")
    m.close()
