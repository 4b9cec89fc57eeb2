#!/usr/bin/env python3
"""The same short run of the native sampler on a DENSE cohort (leapfrog-train units) repeated: identical bits every time?
usage: probe_dense_determinism.py [reps] [chains] [N] [G]; ABD_SAMPLER_UNIT sets the chains per unit; PROBE_GIBBS=0: NUTS only"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
G = int(sys.argv[4]) if len(sys.argv) > 4 else 100
gibbs = os.environ.get("PROBE_GIBBS", "1") == "1"
iters = int(os.environ.get("PROBE_ITERS", "25"))
sc = synthetic.make_cohort(N, G, seed=3)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
th0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
ref, bad = None, 0
for r in range(reps):
    for c in range(C):
        ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
    smp = ctx.sampler(np.arange(C), th0, tune=15, seed=1, gibbs=gibbs)
    th, st = smp.run(iters)
    smp.close()
    if ref is None:
        ref, ref_st = th.copy(), st
    elif not np.array_equal(th, ref):
        bad += 1
        first = np.argwhere((th != ref).any(-1))
        c0, it0 = first[0].tolist()
        print(f"rep {r}: theta differs first at chain {c0}, iteration {it0}; chains affected {sorted(set(first[:, 0].tolist()))}; "
              f"lp {st['lp'][c0, max(0, it0 - 1):it0 + 1]} vs {ref_st['lp'][c0, max(0, it0 - 1):it0 + 1]}; n_steps "
              f"{st['n_steps'][c0, max(0, it0 - 1):it0 + 1]} vs {ref_st['n_steps'][c0, max(0, it0 - 1):it0 + 1]}; gibbs accepted "
              f"{st['gibbs_accepted'][c0, max(0, it0 - 1):it0 + 1]} vs {ref_st['gibbs_accepted'][c0, max(0, it0 - 1):it0 + 1]}", flush=True)
print(f"{bad} of {reps} runs differed (chains {C}, unit {os.environ.get('ABD_SAMPLER_UNIT', 'auto')}, gibbs {gibbs}, {N} x {G}); wait fall-backs {ctx.wait_fallbacks}")
