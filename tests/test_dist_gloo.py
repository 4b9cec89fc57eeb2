"""The N>1 layout on CPU: two gloo ranks shard chains and all-gather their sample blocks (no data-path collective)."""
import os
import socket

import numpy as np
import pytest

from abdpymc_amd.distributed import chain_ids_for_rank, gather_samples, shard_chains


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chains = chain_ids_for_rank(rank, world, 4)
    # each rank "evaluates" its chains: block[c, k, :] = f(global chain id, draw)
    block = np.array([[[cid * 1000 + k + 0.5 * j for j in range(18)] for k in range(5)] for cid in chains])
    allb = gather_samples(block, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, chains, allb))


def test_two_rank_gather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3] and res[1][1] == [4, 5, 6, 7]
    for _, _, allb in res:
        assert allb.shape == (2, 4, 5, 18)
        flat = allb.reshape(8, 5, 18)
        for cid in range(8):
            assert flat[cid, 3, 2] == cid * 1000 + 3 + 1.0


def test_sharding_helpers():
    assert shard_chains(5, 2) == [[0, 2, 4], [1, 3]]
    assert chain_ids_for_rank(3, 8, 4) == [12, 13, 14, 15]
    with pytest.raises(ValueError):
        chain_ids_for_rank(2, 2, 1)
    b = np.arange(6.0).reshape(2, 3)
    np.testing.assert_array_equal(gather_samples(b), b[None])
