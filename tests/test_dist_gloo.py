"""The N>1 layout on CPU: two gloo ranks shard chains and all-gather their sample blocks (no data-path collective)."""
import os
import socket

import numpy as np
import pytest

from abdpymc_amd.distributed import chain_ids_for_rank, gather_results, gather_samples, shard_chains, split_counts


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
    # uneven shards (3 chains over 2 ranks) of a sampler result: padded for the all-gather, trimmed after
    counts = split_counts(3, world)
    first = sum(counts[:rank])
    res = {"p": np.array([[10.0 * (first + c) + k for k in range(4)] for c in range(counts[rank])]),
           "i_raw": np.array([np.full((4, 2, 3), first + c, dtype=np.int8) for c in range(counts[rank])])}
    merged = gather_results(res, counts, dist)
    if rank == 0:  # only the rank that writes the posterior holds the merged arrays, each in its own dtype
        assert merged["p"].shape == (3, 4) and merged["i_raw"].shape == (3, 4, 2, 3) and merged["i_raw"].dtype == np.int8
        assert merged["p"].dtype == np.float64
        assert merged["p"][:, 1].tolist() == [1.0, 11.0, 21.0] and merged["i_raw"][:, 0, 0, 0].tolist() == [0, 1, 2]
    else:
        assert merged is None
    # a THINNED run (abdpymc-infer --thin 3 over 7 draws): the 17 scalars and the statistics keep every draw, the
    # (gap, ind) arrays every third one (draw_index says which), the posterior means have no draw axis -- one gather
    # per array, each with its own shape and dtype; nothing of the size of an unthinned run crosses ranks
    draws, thin = 7, 3
    idx = np.arange(0, draws, thin)
    thinned = {"p": np.array([[100.0 * (first + c) + k for k in range(draws)] for c in range(counts[rank])]),
               "stat_lp": np.array([[-1.0 * (first + c) - k for k in range(draws)] for c in range(counts[rank])]),
               "i": np.array([np.stack([np.full((2, 3), 10 * (first + c) + k, dtype=np.int8) for k in idx]) for c in range(counts[rank])]),
               "ab_n_mu": np.array([np.stack([np.full((2, 3), 0.5 + (first + c) + k) for k in idx]) for c in range(counts[rank])]),
               "draw_index": np.tile(idx, (counts[rank], 1)),
               "mean_i": np.array([np.full((2, 3), 0.25 * (first + c)) for c in range(counts[rank])])}
    bytes_sent = sum(v.nbytes for v in thinned.values())
    merged = gather_results(thinned, counts, dist)
    if rank == 0:
        assert merged["p"].shape == (3, draws) and merged["i"].shape == (3, len(idx), 2, 3) and merged["i"].dtype == np.int8
        assert merged["draw_index"].tolist() == [idx.tolist()] * 3 and merged["mean_i"].shape == (3, 2, 3)
        assert merged["i"][:, :, 0, 0].tolist() == [[10 * c + k for k in idx] for c in range(3)]
        assert merged["ab_n_mu"][2, 1, 1, 2] == 0.5 + 2 + 3 and merged["stat_lp"][1, 6] == -7.0
        # the thinned arrays are draws[::thin] of what an unthinned run would have gathered
        assert bytes_sent < counts[rank] * draws * (2 * 3) * 9 + counts[rank] * (2 * draws + len(idx)) * 8 + counts[rank] * 6 * 8 + 1
    else:
        assert merged is None
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
    assert split_counts(32, 8) == [4] * 8 and split_counts(5, 3) == [2, 2, 1] and split_counts(2, 4) == [1, 1, 0, 0]
    one = {"p": np.ones((2, 3))}
    assert gather_results(one, [2]) is one  # single process: nothing to gather
    assert chain_ids_for_rank(3, 8, 4) == [12, 13, 14, 15]
    with pytest.raises(ValueError):
        chain_ids_for_rank(2, 2, 1)
    b = np.arange(6.0).reshape(2, 3)
    np.testing.assert_array_equal(gather_samples(b), b[None])


def test_slice_individuals():
    from abdpymc_amd import synthetic
    from abdpymc_amd.data import TiterData
    from abdpymc_amd.distributed import slice_individuals

    sc = synthetic.make_cohort(11, 6, seed=2)
    td = TiterData.from_arrays(6, 11, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)
    part = slice_individuals(td, 4, 9)
    assert part.n_inds == 5 and part.n_gaps == 6 and len(part.s) == 5 * 6 and len(part.n) == 5 * 6
    assert part.s.idx_ind.min() == 0 and part.s.idx_ind.max() == 4
    np.testing.assert_array_equal(part.vacs, np.asarray(td.vacs)[4:9])
    keep = (td.n.idx_ind >= 4) & (td.n.idx_ind < 9)
    np.testing.assert_array_equal(part.n.od, td.n.od[keep])
    np.testing.assert_array_equal(part.n.idx_gap, td.n.idx_gap[keep])
