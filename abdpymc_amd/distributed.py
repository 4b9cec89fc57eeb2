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
