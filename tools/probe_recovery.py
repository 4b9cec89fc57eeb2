#!/usr/bin/env python3
"""Parameter recovery on a simulated cohort with the native sampler (development probe)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic
from abdpymc_amd.data import TiterData
from abdpymc_amd.model import model
from abdpymc_amd.sampler import sample

N, G = int(sys.argv[1]), int(sys.argv[2])
tune, draws, chains = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dense = len(sys.argv) > 6 and sys.argv[6] == "dense"
sc = synthetic.make_cohort(N, G, seed=77)
td = TiterData.from_arrays(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)
m = model(td, n_chains=chains)
t0 = time.perf_counter()
res = sample(m, tune, draws, chains=chains, seed=3, record_deterministics=False, record_discrete=False, dense_metric=dense)
print(f"dense={dense} n_steps {res['stat_n_steps'].mean():.1f}", end=" ")
print(f"{time.perf_counter() - t0:.2f} s; depth {res['stat_tree_depth'].mean():.2f}; div {int(res['stat_diverging'].sum())}; gibbs acc {res['stat_gibbs_accept'].mean():.4f}")
for k, v in synthetic.TRUTH.items():
    d = res[k]
    print(f"{k:12s} truth {v:6.3f}  post {d.mean():7.4f} +- {d.std():.4f}   per chain {np.round(d.mean(1), 4)}")
print("p", res["p"].mean(), "p_waner", res["ab_s_p_waner"].mean())
mi = res["mean_i"].mean(0)
it = sc.i_true.astype(bool)
print(f"P(i | true infection) {mi[it].mean():.3f}   P(i | none) {mi[~it].mean():.5f}   n_true {it.sum()}  expected n_inf {mi.sum():.1f}")
m.close()
