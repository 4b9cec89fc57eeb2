"""
A PyMC-shaped step over the device sweep.

``pm.sample`` assigns ``BinaryGibbsMetropolis`` to ``[i_raw, ab_s_waner]`` (reference call site abd.py:922).  Its
``astep`` flips one raveled bit at a time and asks the compiled ``logp`` for the JOINT log-probability after every flip --
about 0.8 (G N + N) full evaluations per draw (1.6 M at BASELINE config 3), the loop that dominates ``abdpymc-infer``.
``abd_gibbs_sweep`` is the same sweep in one device launch (abdpymc_amd/csrc/abd_gibbs*.hpp: a flip only changes its own
individual's terms, so individuals sweep independently and in parallel).  This module wraps it in the call shape of a PyMC
step:

* :class:`GibbsSweepStep` -- no PyMC needed: ``astep(point) -> (new_point, stats)``; uploads the point's discrete state
  only if the device does not already hold it (``DiscreteMirror``), runs ONE device sweep at the point's continuous
  values and returns the point with the new ``i_raw`` / ``ab_s_waner``.  Tested on the GPU against ``Context.gibbs_sweep``.
* ``DeviceBinaryGibbs`` -- the ``pm.step_methods`` subclass a maintainer of the reference hands to
  ``pm.sample(step=[nuts, DeviceBinaryGibbs(...)])``.  UNVERIFIED-OFFLINE (PyMC is not installed where this was
  written); import-guarded, a thin shell around :class:`GibbsSweepStep`.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from ._native import DiscreteMirror
from .model import AbdModel

try:  # pragma: no cover - not importable offline
    from pymc.step_methods.arraystep import BlockedStep

    HAVE_PYMC = True
except ImportError:  # pragma: no cover
    HAVE_PYMC = False
    BlockedStep = object


class GibbsSweepStep:
    """One binary Gibbs-Metropolis sweep of ``[i_raw, ab_s_waner]`` per call, on the device.

    Randomness: Philox keyed by ``(seed, sweep number)`` -- the sweep number counts the calls, so a chain's sweeps are
    reproducible and independent of what other chain slots do (``abd_hip.h``: abd_gibbs_sweep).  Every CHAIN needs its own
    seed: ``pm.sample`` copies a step method into each chain (or worker process), and copies that share a seed would propose
    with the same random numbers.  ``seed=None`` (the default) therefore draws the seed when the FIRST sweep is asked for --
    after the copy was made -- from the generator PyMC hands the chain's step (``set_rng``) or, failing that, from the
    operating system; pass an int (a different one per chain) for a reproducible run.
    """

    def __init__(self, model: AbdModel, chain: int = 0, seed=None):
        self.model, self.chain = model, int(chain)
        self.seed = None if seed is None else int(seed)
        self.mirror = DiscreteMirror(model.ctx, self.chain)
        self.n_sweeps = 0
        self.accepted = self.proposed = 0

    def set_rng(self, rng) -> None:
        """PyMC >= 5.x gives every chain's step methods a generator of their own: the sweep's seed comes from it."""
        self.seed = int(np.random.default_rng(rng).integers(0, 2 ** 63 - 1))

    def _seed(self) -> int:
        if self.seed is None:
            self.seed = int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array([1, 2 ** 31], dtype=np.uint64))
        return self.seed

    def astep(self, point: Dict[str, np.ndarray]) -> Tuple[Dict[str, np.ndarray], Dict[str, int]]:
        ctx = self.model.ctx
        i_raw, waner = np.asarray(point["i_raw"]), np.asarray(point["ab_s_waner"])
        self.mirror.update(i_raw, waner)  # nothing moves if the device already holds this state
        theta = self.model.ravel(point)
        acc, prop = ctx.gibbs_sweep([self.chain], theta[None, :], seed=self._seed(), sweep=self.n_sweeps)
        self.n_sweeps += 1
        new_i, new_w = ctx.get_discrete(self.chain)
        # the device holds exactly what is handed back: the next call with this state uploads nothing
        self.mirror.adopt(new_i.astype(i_raw.dtype, copy=False), new_w.astype(waner.dtype, copy=False))
        out = dict(point)
        out["i_raw"] = new_i.astype(i_raw.dtype)
        out["ab_s_waner"] = new_w.astype(waner.dtype)
        self.accepted += int(acc[0])
        self.proposed += int(prop[0])
        return out, {"accepted": int(acc[0]), "proposed": int(prop[0])}


if HAVE_PYMC:  # pragma: no cover

    class DeviceBinaryGibbs(BlockedStep):
        """``pm.sample(step=[pm.NUTS(cont, logp_dlogp_func=gpu.logp_dlogp_function()), DeviceBinaryGibbs(disc, gpu)])``:
        replaces ``pm.BinaryGibbsMetropolis(disc)`` -- one device launch per draw instead of one ``logp`` call per bit."""

        name = "device_binary_gibbs"
        generates_stats = True
        stats_dtypes_shapes = {"accepted": (np.int64, []), "proposed": (np.int64, [])}
        stats_dtypes = [{"accepted": np.int64, "proposed": np.int64}]

        def __init__(self, vars, gpu_model: AbdModel, chain: int = 0, seed=None, model=None):
            self.vars = list(vars)
            self.core = GibbsSweepStep(gpu_model, chain=chain, seed=seed)

        def set_rng(self, rng):
            self.core.set_rng(rng)

        def step(self, point):
            new_point, stats = self.core.astep(point)
            return new_point, [stats]
