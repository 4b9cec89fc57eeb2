import sys, time, numpy as np
sys.path.insert(0, '.')
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context
N, G, C, K = 10000, 200, 4, 20
sc = synthetic.make_cohort(N, G)
ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
for c in range(C):
    ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
th = np.stack([synthetic.make_thetas(G, K, c) for c in range(C)], axis=1)
chains = np.arange(C, dtype=np.int32)
def region():
    t0 = time.perf_counter()
    for k in range(K):
        ctx.enqueue(k, chains, th[k])
    ctx.wait()
    ctx.fetch_many(np.arange(K), C)
    return time.perf_counter() - t0
for _ in range(200): region()
for mode in (0, 2, 0, 2):
    ctx.kernel_timing(mode)
    ctx.kernel_time(reset=True)
    ts = [region() for _ in range(200)]
    ms, n = ctx.kernel_time(reset=True)
    print(f"mode {mode}: host median region {np.median(ts)*1e6:.1f} us = {np.median(ts)*1e6/K:.2f} us/step; device window {ms*1e3/max(n,1):.2f} us/launch over {n} launches")
ctx.kernel_timing(0)
