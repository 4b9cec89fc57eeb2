"""
One chain over several processes: the cohort sharded by individual (abdpymc_amd.distributed.IndividualShards).
Two ranks share this box's one GPU (gloo carries the 18-double all-reduce); the put-together logp / gradient must
equal the unsharded evaluation and the sharded Gibbs sweep must be the unsharded one bit for bit.
"""
import os
import socket

import numpy as np
import pytest

from abdpymc_amd import synthetic
from abdpymc_amd.data import TiterData

pytestmark = pytest.mark.gpu


def _cohort():
    sc = synthetic.make_cohort(301, 70, seed=41)  # odd count: shards of 151 and 150
    return TiterData.from_arrays(70, 301, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)


def _state(seed=3):
    rng = np.random.default_rng(seed)
    i_raw = (rng.random((70, 301)) < 0.03).astype(np.int8)
    w = (rng.random(301) < 0.5).astype(np.int8)
    theta = synthetic.theta_init(70) + 0.2 * rng.standard_normal(17)
    return theta, i_raw, w


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from abdpymc_amd.distributed import IndividualShards

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = IndividualShards(_cohort(), dist, splits=(30,), n_chains=1, device=0)
    theta, i_raw, w = _state()
    sh.set_discrete(0, i_raw, w)
    lp, g = sh.logp_dlogp(0, theta)
    acc, prop = sh.gibbs_sweep([0], theta[None], seed=77, sweep=5)
    i_part, w_part = sh.get_discrete(0)
    lp2, g2 = sh.logp_dlogp(0, theta)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sh.j0, sh.j1, lp, g, int(acc[0]), int(prop[0]), i_part, w_part, lp2, g2))
    sh.close()


def test_two_shards_equal_the_whole():
    import torch.multiprocessing as mp

    from abdpymc_amd.model import model

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # the unsharded answer
    m = model(_cohort(), splits=(30,), n_chains=1)
    theta, i_raw, w = _state()
    m.ctx.set_discrete(0, i_raw, w)
    lp, g = m.ctx.logp_dlogp(0, theta)
    acc, prop = m.ctx.gibbs_sweep([0], theta[None], seed=77, sweep=5)
    i_new, w_new = m.ctx.get_discrete(0)
    lp2, g2 = m.ctx.logp_dlogp(0, theta)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 151, 151, 301)
    for r in res:
        assert abs(r[3] - lp) <= 1e-12 * abs(lp)
        np.testing.assert_allclose(r[4], g, rtol=0, atol=1e-11 * np.abs(g).max())
        np.testing.assert_array_equal(r[7], i_new[:, r[1]:r[2]])  # the sweep: same bits as the unsharded one
        np.testing.assert_array_equal(r[8], w_new[r[1]:r[2]])
        assert abs(r[9] - lp2) <= 1e-12 * abs(lp2)
        np.testing.assert_allclose(r[10], g2, rtol=0, atol=1e-11 * np.abs(g2).max())
    assert res[0][5] + res[1][5] == int(acc[0]) and res[0][6] + res[1][6] == int(prop[0])
    assert not np.array_equal(i_new, i_raw)  # the sweep did something
    m.close()
