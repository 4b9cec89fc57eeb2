#!/usr/bin/env python3
"""Kernel-time sweep over launch configurations (blocks, chains per wave) on one GPU.  Dev tool."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd._native import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-inds", type=int, default=10000)
ap.add_argument("--n-gaps", type=int, default=200)
ap.add_argument("--chains", type=int, default=4)
ap.add_argument("--storage", default="f64")
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--blocks", default="0", help="dense: segment lengths (0 = auto); sparse: grid blocks")
ap.add_argument("--cpw", default="4,2,1")
args = ap.parse_args()

G, N, C = args.n_gaps, args.n_inds, args.chains
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C, storage=args.storage)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th = np.stack([synthetic.make_thetas(G, args.iters + 5, c) for c in range(C)], axis=1)
chains = np.arange(C, dtype=np.int32)
alg = ctx.algorithmic_bytes(C)
print(f"# {ctx.device_name}  N={N} G={G} chains={C} storage={args.storage} alg_bytes/launch={alg}")
for cpw in [int(x) for x in args.cpw.split(",")]:
    for blocks in [int(x) for x in args.blocks.split(",")]:
        ctx.set_launch_config(blocks, cpw)
        for k in range(5):
            ctx.enqueue(k, chains, th[k])
        ctx.wait()
        ctx.kernel_timing(True)
        ctx.kernel_time(reset=True)
        for k in range(args.iters):
            ctx.enqueue(k, chains, th[5 + k])
        ms, n = ctx.kernel_time(reset=True)
        ctx.kernel_timing(False)
        us = ms / n * 1e3
        print(f"cpw={cpw} blocks={blocks:5d} launches={n:4d} kernel_us={us:9.2f} per_step_us={us * n / args.iters:9.2f} "
              f"GB/s={alg / (ms / args.iters * 1e-3) / 1e9:8.1f}", flush=True)
