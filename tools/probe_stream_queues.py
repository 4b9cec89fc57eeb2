import sys, numpy as np
sys.path.insert(0,'.')
from abdpymc_amd import synthetic
from abdpymc_amd._native import Context
sc = synthetic.make_cohort(1000, 60)
for rep in range(3):
    ctx = Context(60, 1000, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=4)
    print("queues of the 8 streams:", ctx.stream_queues())
    ctx.close()
import torch
torch.zeros(4, device="cuda")
ctx = Context(60, 1000, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=4)
print("after torch touched the GPU:", ctx.stream_queues())
