"""
Multi-GPU layout of the path: chains are independent (own theta, own i_raw / waner), the cohort panels are
replicated read-only on every GPU, so chains are sharded across ranks -- one process per GPU -- with NO
collective on the data path.  The only exchange is a gather of each rank's sample block
(chains_per_rank x draws x width doubles) at the end (or every M draws): ``torch.distributed.all_gather``,
which is RCCL over xGMI with backend "nccl" and runs unchanged on CPU tensors with "gloo".

The reference has no communication backend of its own: PyMC moves each chain's draws from its worker process
to the parent over multiprocessing pipes (pm.sample(cores=...), abd.py:922).
"""
from __future__ import annotations

from typing import List

import numpy as np


def chain_ids_for_rank(rank: int, world: int, chains_per_rank: int) -> List[int]:
    """Global chain ids owned by ``rank`` (contiguous blocks: rank r owns [r*c, (r+1)*c))."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    return list(range(rank * chains_per_rank, (rank + 1) * chains_per_rank))


def shard_chains(n_chains: int, world: int) -> List[List[int]]:
    """Round-robin split of n_chains over world ranks when they do not divide evenly."""
    return [list(range(r, n_chains, world)) for r in range(world)]


def gather_samples(block: np.ndarray, dist=None, device=None) -> np.ndarray:
    """
    All-gather equally shaped per-rank sample blocks; returns (world, *block.shape) on every rank.
    ``dist`` is an initialised torch.distributed module (None = single process).
    """
    block = np.ascontiguousarray(block, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return block[None]
    import torch

    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def split_counts(n_chains: int, world: int) -> List[int]:
    """Chains per rank for contiguous shards: the first ``n_chains % world`` ranks get one more."""
    base, extra = divmod(n_chains, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def gather_results(res: dict, counts: List[int], dist=None, device=None, dst: int = 0):
    """
    Concatenate per-rank result dicts (arrays with a leading local-chain axis) along the chain axis, in rank
    order, ON RANK ``dst`` ONLY (the rank that writes the posterior); the other ranks get ``None``.  Every array
    travels in its own dtype (the int8 ``i_raw`` / ``i`` panels of a recorded run are the bulk of the bytes: as
    float64 all-gathered to every rank, as before, 1 000 draws of config 3 were 16 GB per chain per rank).
    Shards may differ in size by one chain: blocks are padded to the largest shard and trimmed afterwards.
    """
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return res
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    cmax = max(counts)
    out = {} if rank == dst else None
    for key in sorted(res):
        a = np.ascontiguousarray(res[key])
        if a.ndim == 0 or a.shape[0] != counts[rank]:
            raise ValueError(f"{key}: leading axis {a.shape} is not this rank's chain count {counts[rank]}")
        pad = np.zeros((cmax,) + a.shape[1:], dtype=a.dtype)
        pad[: a.shape[0]] = a
        t = torch.from_numpy(pad)
        if device is not None:
            t = t.to(device)
        parts = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, parts, dst=dst)
        if rank == dst:
            out[key] = np.concatenate([parts[r][: counts[r]].cpu().numpy() for r in range(world)])
    return out


def init_from_env():
    """
    One process per GPU under ``torch.distributed.run``: returns (dist, rank, world, local_rank), or
    (None, 0, 1, 0) when not launched that way.  Backend "nccl" (RCCL over xGMI) when every rank has its own
    GPU, "gloo" otherwise or when ABD_DIST_BACKEND says so.
    """
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1, 0
    import torch
    import torch.distributed as dist

    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ABD_DIST_BACKEND") or ("nccl" if torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo")
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank, world, local


# ---------------------------------------------------------------------------------------------------
# One chain over several GPUs: the cohort sharded by individual (second way)
# ---------------------------------------------------------------------------------------------------


def slice_individuals(data, j0: int, j1: int):
    """The TiterData of individuals [j0, j1): their observations (re-indexed from 0), vaccinations and PCR+ rows."""
    from .data import TiterData

    def obs(a):
        keep = (a.idx_ind >= j0) & (a.idx_ind < j1)
        return a.idx_gap[keep], a.idx_ind[keep] - j0, a.log_dilution[keep], a.od[keep]

    return TiterData.from_arrays(data.n_gaps, j1 - j0, obs(data.s), obs(data.n), np.asarray(data.vacs)[j0:j1],
                                 np.asarray(data.pcrpos)[j0:j1], t0=data.t0)


class IndividualShards:
    """
    This process's slice of a cohort sharded by individual over ``dist``'s ranks (one per GPU), with the joint logp
    and its gradient put back together by ONE all-reduce of 18 doubles per evaluation:
    ``sum over ranks of logp_r - (world - 1) * theta_prior`` (every rank's own logp carries the theta-only terms
    once).  The Gibbs sweep touches only a rank's own individuals and needs no exchange; it draws the random
    numbers of each individual's global index, so the sharded sweep is the unsharded one bit for bit.
    """

    def __init__(self, data, dist=None, splits=None, ignore_pcrpos=False, n_chains=1, device=-1, all_reduce_device=None):
        from .model import AbdModel

        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        counts = split_counts(data.n_inds, self.world)
        self.j0 = sum(counts[: self.rank])
        self.j1 = self.j0 + counts[self.rank]
        if self.j1 == self.j0:
            raise ValueError(f"{data.n_inds} individuals cannot be sharded over {self.world} processes")
        self.model = AbdModel(slice_individuals(data, self.j0, self.j1), splits=splits, ignore_pcrpos=ignore_pcrpos,
                              n_chains=n_chains, device=device)
        self.model.ctx.set_individual_offset(self.j0)
        self._device = all_reduce_device

    def set_discrete(self, chain: int, i_raw, waner):
        """i_raw (G, N) and waner (N,) of the WHOLE cohort; this rank keeps its columns."""
        self.model.ctx.set_discrete(chain, np.asarray(i_raw)[:, self.j0:self.j1], np.asarray(waner)[self.j0:self.j1])

    def get_discrete(self, chain: int):
        """This rank's columns: i_raw (G, j1 - j0), waner (j1 - j0,)."""
        return self.model.ctx.get_discrete(chain)

    def logp_dlogp(self, chain: int, theta):
        lp, g = self.model.ctx.logp_dlogp(chain, theta)
        if self.dist is None:
            return lp, g
        import torch

        buf = torch.from_numpy(np.concatenate([[lp], g]))
        if self._device is not None:
            buf = buf.to(self._device)
        self.dist.all_reduce(buf)  # sum: RCCL over xGMI with backend "nccl"
        tot = buf.cpu().numpy()
        plp, pg = self.model.ctx.theta_prior(theta)
        return float(tot[0] - (self.world - 1) * plp), tot[1:] - (self.world - 1) * pg

    def gibbs_sweep(self, chains, theta, seed: int, sweep: int):
        """Local sweep of this rank's individuals -> (accepted, proposed) of this rank."""
        return self.model.ctx.gibbs_sweep(chains, theta, seed=seed, sweep=sweep)

    def close(self):
        self.model.close()
