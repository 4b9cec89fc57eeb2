#!/usr/bin/env python3
"""End-to-end iterations per second of the compound sampler ([NUTS; Gibbs sweep; record]) on one GPU.

usage: bench_sampler.py [default|c2|c3] [--chains C] [--tune T] [--draws D] [--no-record]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import sampler, synthetic  # noqa: E402
from abdpymc_amd.data import TiterData  # noqa: E402
from abdpymc_amd.model import model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cohort", nargs="?", default="default")
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--tune", type=int, default=100)
    ap.add_argument("--draws", type=int, default=100)
    ap.add_argument("--no-record", action="store_true")
    ap.add_argument("--python", action="store_true", help="the Python driver instead of the native one")
    ap.add_argument("--dense", action="store_true", help="dense metric")
    a = ap.parse_args()
    if a.cohort == "default":
        from tests.test_data_loader import default_cohort

        td = default_cohort(os.path.join(ROOT, "tests", "golden"))
        m = model(td, splits=(14, 20), n_chains=a.chains)
    else:
        n, g = {"c2": (1000, 60), "c3": (10000, 200)}[a.cohort]
        sc = synthetic.make_cohort(n, g)
        td = TiterData.from_arrays(g, n, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)
        m = model(td, n_chains=a.chains)
    t0 = time.perf_counter()
    kw = dict(native=not a.python) if "native" in sampler.sample.__code__.co_varnames else {}
    if a.dense:
        kw["dense_metric"] = True
    res = sampler.sample(m, a.tune, a.draws, chains=a.chains, seed=1, record_deterministics=not a.no_record,
                         record_discrete=not a.no_record, **kw)
    dt = time.perf_counter() - t0
    iters = a.tune + a.draws
    print(f"{a.cohort}: {a.chains} chains x {iters} iterations in {dt:.2f} s = {a.chains * iters / dt:.1f} chain-iterations/s; "
          f"mean tree depth {res['stat_tree_depth'].mean():.2f}, n_steps {res['stat_n_steps'].mean():.1f}, "
          f"accept {res['stat_mean_tree_accept'].mean():.2f}, gibbs accept {res['stat_gibbs_accept'].mean():.3f}, "
          f"grad evals {int(np.sum(res['n_grad_evals']))}")
    m.close()


if __name__ == "__main__":
    main()
