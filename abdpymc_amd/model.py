"""
The antibody-dynamics model behind the two callables PyMC's step methods consume.

``model(data, splits=None, ignore_pcrpos=False)`` mirrors ``abdpymc.model`` (reference abd.py:396-442): same
arguments, same validation, same 19 value variables in the same order with PyMC v5's transformed names.
Instead of a ``pm.Model`` whose logp PyTensor compiles, it returns an :class:`AbdModel` whose

* ``compile_logp()``          -> ``logp_fn(point) -> float64``      (Model.compile_logp())
* ``logp_dlogp_function()``   -> ``fn(q[17]) -> (logp, grad[17])``  (Model.logp_dlogp_function())
* ``deterministics(point)``   -> ``{"i", "ab_n_mu", "ab_s_mu"}``    (pm.Deterministic, abd.py:649/667, 341, 389-391)

run on the MI355X through the C ABI.  There is no CPU path: constructing an AbdModel needs the HIP library
and a GPU.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np

from . import _native
from .data import TiterData, check_splits

GAP_IND = ("gap", "ind")

# continuous value variables, creation order (abd.py:424, 329-340, 367-388, 464-467); PyMC v5 names
THETA_NAMES = (
    "p_logodds__",
    "ab_n_perm_log__",
    "ab_n_temp_log__",
    "ab_n_rho_logodds__",
    "ab_n_init",
    "ab_s_perm_log__",
    "ab_s_rho_logodds__",
    "ab_s_p_waner_logodds__",
    "ab_s_tempinf_log__",
    "ab_s_tempvac_log__",
    "ab_s_init",
    "it_n_b",
    "it_n_d",
    "it_n_sigma_log__",
    "it_s_b",
    "it_s_d",
    "it_s_sigma_log__",
)
DISCRETE_NAMES = ("i_raw", "ab_s_waner")
# all 19 in creation order
VALUE_VAR_NAMES = (
    ("p_logodds__", "i_raw")
    + THETA_NAMES[1:8]
    + ("ab_s_waner",)
    + THETA_NAMES[8:]
)
# constrained names and their backward transforms
_TRANSFORMS = {
    "p_logodds__": ("p", "logodds"),
    "ab_n_perm_log__": ("ab_n_perm", "log"),
    "ab_n_temp_log__": ("ab_n_temp", "log"),
    "ab_n_rho_logodds__": ("ab_n_rho", "logodds"),
    "ab_n_init": ("ab_n_init", None),
    "ab_s_perm_log__": ("ab_s_perm", "log"),
    "ab_s_rho_logodds__": ("ab_s_rho", "logodds"),
    "ab_s_p_waner_logodds__": ("ab_s_p_waner", "logodds"),
    "ab_s_tempinf_log__": ("ab_s_tempinf", "log"),
    "ab_s_tempvac_log__": ("ab_s_tempvac", "log"),
    "ab_s_init": ("ab_s_init", None),
    "it_n_b": ("it_n_b", None),
    "it_n_d": ("it_n_d", None),
    "it_n_sigma_log__": ("it_n_sigma", "log"),
    "it_s_b": ("it_s_b", None),
    "it_s_d": ("it_s_d", None),
    "it_s_sigma_log__": ("it_s_sigma", "log"),
}


def constrain(theta: np.ndarray) -> Dict[str, np.ndarray]:
    """theta (..., 17) on the unconstrained scale -> constrained free variables by name."""
    theta = np.asarray(theta, dtype=float)
    out = {}
    for k, name in enumerate(THETA_NAMES):
        cname, tr = _TRANSFORMS[name]
        v = theta[..., k]
        if tr == "log":
            v = np.exp(v)
        elif tr == "logodds":
            v = 1.0 / (1.0 + np.exp(-v))
        out[cname] = v
    return out


class ValueGradFunction:
    """
    Call shape of PyMC's ``ValueGradFunction``: ``fn(q) -> (logp, dlogp)`` for the raveled continuous
    variables, the discrete ones set beforehand with ``set_extra_values(point)``.
    """

    dtype = "float64"
    profile = None

    def __init__(self, model: "AbdModel", chain: int = 0):
        self._model = model
        self._chain = chain
        self._extra_are_set = False
        self._extra_vars_shared: Dict[str, np.ndarray] = {}
        self._mirror = _native.DiscreteMirror(model.ctx, chain)

    def set_extra_values(self, point: Dict[str, np.ndarray]) -> None:
        i_raw = np.asarray(point["i_raw"])
        waner = np.asarray(point["ab_s_waner"])
        # uploads what changed since this callable last looked -- unless someone else (the point function of
        # compile_logp(), a device sweep) has rewritten the slot meanwhile: then the whole state goes up again
        self._mirror.update(i_raw, waner)
        self._extra_vars_shared = {"i_raw": i_raw.copy(), "ab_s_waner": waner.copy()}
        self._extra_are_set = True

    def get_extra_values(self) -> Dict[str, np.ndarray]:
        if not self._extra_are_set:
            raise ValueError("Extra values are not set.")
        return dict(self._extra_vars_shared)

    def __call__(self, q, *, extra_vars=None):
        if extra_vars is not None:
            self.set_extra_values(extra_vars)
        if not self._extra_are_set:
            raise ValueError("Extra values are not set.")
        q = np.asarray(getattr(q, "data", q), dtype=np.float64)
        return self._model.ctx.logp_dlogp(self._chain, q)


class AbdModel:
    """What ``abd.model()`` builds, with the joint logp evaluated on the GPU."""

    def __init__(
        self,
        data: TiterData,
        splits: Optional[Sequence[int]] = None,
        ignore_pcrpos: bool = False,
        n_chains: int = 1,
        storage: str = "f64",
        device: int = -1,
    ):
        check_splits(splits=splits, data=data)  # abd.py:410
        if splits is not None and len(splits) > 2:
            raise NotImplementedError("only implemented 1-3 time chunks (0-2 splits)")  # abd.py:882
        self.data = data
        self.splits = tuple(splits) if splits else ()
        self.ignore_pcrpos = bool(ignore_pcrpos)
        self.n_gaps, self.n_inds = data.n_gaps, data.n_inds
        self.n_chains = int(n_chains)
        self.coords = data.coords
        if np.asarray(data.vacs).shape != (self.n_inds, self.n_gaps):
            raise ValueError(
                f"vacs / pcrpos shape {np.asarray(data.vacs).shape} != (n_inds, n_gaps) = {(self.n_inds, self.n_gaps)}"
            )
        self.ctx = _native.Context(
            self.n_gaps,
            self.n_inds,
            data.s.obs,
            data.n.obs,
            data.vacs,
            None if ignore_pcrpos else data.pcrpos,  # abd.py:416-418
            splits=self.splits,
            n_chains=self.n_chains,
            storage=storage,
            device=device,
        )

    # -- variables ------------------------------------------------------------------------------
    @property
    def value_vars(self):
        return list(VALUE_VAR_NAMES)

    @property
    def continuous_value_vars(self):
        return list(THETA_NAMES)

    @property
    def discrete_value_vars(self):
        return list(DISCRETE_NAMES)

    def shapes(self) -> Dict[str, tuple]:
        sh = {n: () for n in THETA_NAMES}
        sh["i_raw"] = (self.n_gaps, self.n_inds)
        sh["ab_s_waner"] = (self.n_inds,)
        return sh

    def initial_point(self) -> Dict[str, np.ndarray]:
        """PyMC's default: prior means on the transformed scale; Bernoulli support point p<0.5 -> 0."""

        def logit(x):
            return math.log(x / (1.0 - x))

        pt = {
            "p_logodds__": logit(1.0 / self.n_gaps),  # mean of Beta(1, G-1)
            "ab_n_perm_log__": math.log(2.0),
            "ab_n_temp_log__": math.log(1.0),
            "ab_n_rho_logodds__": logit(10.0 / 11.0),
            "ab_n_init": -2.0,
            "ab_s_perm_log__": math.log(2.0),
            "ab_s_rho_logodds__": logit(10.0 / 11.0),
            "ab_s_p_waner_logodds__": 0.0,
            "ab_s_tempinf_log__": math.log(1.0),
            "ab_s_tempvac_log__": math.log(1.0),
            "ab_s_init": -2.0,
            "it_n_b": -1.0,
            "it_n_d": 2.0,
            "it_n_sigma_log__": 0.0,
            "it_s_b": -1.0,
            "it_s_d": 2.0,
            "it_s_sigma_log__": 0.0,
        }
        out = {k: np.asarray(v, dtype=np.float64) for k, v in pt.items()}
        out["i_raw"] = np.zeros((self.n_gaps, self.n_inds), dtype=np.int64)
        out["ab_s_waner"] = np.ones(self.n_inds, dtype=np.int64)  # p_waner starts at 0.5 -> support point 1
        return out

    @staticmethod
    def ravel(point: Dict[str, np.ndarray]) -> np.ndarray:
        return np.array([float(np.asarray(point[n])) for n in THETA_NAMES], dtype=np.float64)

    @staticmethod
    def unravel(q: np.ndarray) -> Dict[str, np.ndarray]:
        return {n: np.asarray(q[k], dtype=np.float64) for k, n in enumerate(THETA_NAMES)}

    # -- the two callables --------------------------------------------------------------------------
    def compile_logp(self, chain: int = 0):
        """``logp_fn(point: dict) -> float64``: the PointFunc BinaryGibbsMetropolis calls once per flipped bit."""
        ctx = self.ctx
        mirror = _native.DiscreteMirror(ctx, chain)

        def logp_fn(point: Dict[str, np.ndarray]) -> np.float64:
            # a Gibbs step changes one raveled bit between calls: the mirror flips it on the device instead of
            # re-uploading the panel, and re-uploads when anything else has touched the slot since (the slot's
            # generation counter: ValueGradFunction.set_extra_values, gibbs_sweep, the native sampler)
            mirror.update(point["i_raw"], point["ab_s_waner"])
            return np.float64(ctx.logp(chain, self.ravel(point)))

        logp_fn.invalidate = mirror.invalidate
        logp_fn.mirror = mirror
        return logp_fn

    def logp_dlogp_function(self, chain: int = 0) -> ValueGradFunction:
        return ValueGradFunction(self, chain)

    def deterministics(self, point: Dict[str, np.ndarray], chain: int = 0) -> Dict[str, np.ndarray]:
        self.ctx.set_discrete(chain, np.asarray(point["i_raw"]), np.asarray(point["ab_s_waner"]))
        i, mun, mus = self.ctx.deterministics(chain, self.ravel(point))
        return {"i": i, "ab_n_mu": mun, "ab_s_mu": mus}

    def close(self):
        self.ctx.close()


def model(
    data: TiterData,
    splits: Optional[Sequence[int]] = None,
    ignore_pcrpos: bool = False,
    **kwds,
) -> AbdModel:
    """Set up an antibody dynamics model (same signature as the reference, abd.py:396-400)."""
    return AbdModel(data, splits=splits, ignore_pcrpos=ignore_pcrpos, **kwds)
