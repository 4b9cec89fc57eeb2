#!/usr/bin/env python3
"""CPU baselines of SURVEY 8(d) on the box's host cores (no GPU involved):
 B0  the reference's algorithm restated literally in NumPy: dense (G, G, N) temp responses, forward logp only
     (the reference gets its gradient from reverse-mode autodiff over the same graph), 1 core
 B1  the C restatement (recurrence + analytic gradient, oracle/abd_oracle.c), logp + gradient, 1 core and OpenMP
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from oracle import abd_oracle as O
from oracle import c_oracle
from tests.helpers import oracle_cohort_from_synth


def rate(fn, budget=6.0, min_n=2):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if (el > budget and n >= min_n) or n >= 5000:
            return n / el


threads = max(1, min(c_oracle.max_threads(), len(os.sched_getaffinity(0)), 16))
for name, (N, G) in (("config 2 (1000 x 60)", (1000, 60)), ("config 3 (10000 x 200)", (10000, 200))):
    coh = oracle_cohort_from_synth(synthetic.make_cohort(N, G))
    i_raw, w = synthetic.make_chain_state(N, G, 0)
    theta = synthetic.make_thetas(G, 1, 0)[0]
    co = c_oracle.COracle(coh, None)
    b1_1 = rate(lambda: co.logp_dlogp(theta, i_raw, w, nthreads=1))
    b1_n = rate(lambda: co.logp_dlogp(theta, i_raw, w, nthreads=threads))
    line = f"{name}: B1 C port logp+grad {b1_1:8.2f} evals/s on 1 core, {b1_n:8.2f} on {threads} threads"
    if N * G * G * 8 < 2e9:
        b0 = rate(lambda: O.joint_logp(theta, i_raw, w, coh, dense=True), budget=10.0)
        line += f"; B0 NumPy dense (G,G,N) forward-only {b0:7.2f} evals/s on 1 core"
    else:
        t0 = time.perf_counter()
        O.joint_logp(theta, i_raw, w, coh, dense=True)
        line += f"; B0 NumPy dense (G,G,N) forward-only: {time.perf_counter() - t0:.1f} s per evaluation (3.2 GB temporaries)"
    print(line, flush=True)
