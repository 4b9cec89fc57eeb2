"""
PyTensor / PyMC adapter: the joint data term as an ``Op`` so the reference's own ``pm.sample`` can drive the
HIP library.  UNVERIFIED-OFFLINE: PyMC / PyTensor are not installed in the build container or on the GPU box,
so this module is written against their documented API (PyTensor 2.x ``Op.make_node/perform/grad``,
PyMC v5 ``pm.Potential``) and is import-guarded; nothing else in the package depends on it.

``pymc_model(data, splits, ignore_pcrpos)`` declares the same 19 random variables with the same names and
priors as the reference (abd.py:424-427, 329-340, 367-388, 464-467) -- so step assignment, transforms and the
names in the InferenceData match -- and attaches the likelihood of the two sigmoid panels (abd.py:459-469)
with ``pm.Potential``.  The Op is picklable (multiprocess chains) and does not touch HIP until its first
``perform`` in the worker process.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not importable offline
    import pymc as pm
    import pytensor.tensor as pt
    from pytensor.graph.basic import Apply
    from pytensor.graph.op import Op

    HAVE_PYMC = True
except ImportError:  # pragma: no cover
    HAVE_PYMC = False
    Op = object


class _Lazy:
    """Holds what is needed to rebuild the native context in whichever process first evaluates the Op."""

    def __init__(self, data, splits, ignore_pcrpos):
        self.data, self.splits, self.ignore_pcrpos = data, splits, ignore_pcrpos
        self._model = None
        self._mirror = None

    def model(self):
        if self._model is None:
            from .model import AbdModel

            self._model = AbdModel(self.data, self.splits, self.ignore_pcrpos, n_chains=1)
        return self._model

    def mirror(self):
        """Host mirror of chain slot 0's discrete state: NUTS leapfrogs never change it, a Gibbs proposal changes
        one bit -- ``perform`` uploads only the difference (see _native.DiscreteMirror)."""
        if self._mirror is None:
            from ._native import DiscreteMirror

            self._mirror = DiscreteMirror(self.model().ctx, 0)
        return self._mirror

    def loglik_dlogp_constrained(self, params, i_raw, waner):
        """What ``AbdDataLogp.perform`` computes: the data term and its gradient w.r.t. the 13 constrained parameters."""
        m = self.model()
        self.mirror().update(np.asarray(i_raw), np.asarray(waner))
        ll, g = m.ctx.loglik_dlogp(0, _theta_from_constrained(params, m.n_gaps))  # data term only; PyMC keeps the priors
        return ll, _grad_to_constrained(g, params)

    def __getstate__(self):
        return dict(data=self.data, splits=self.splits, ignore_pcrpos=self.ignore_pcrpos, _model=None, _mirror=None)


if HAVE_PYMC:  # pragma: no cover

    class AbdDataLogp(Op):
        """(theta13, i_raw, waner) -> [data log-likelihood, d/d theta13]: the part of the joint logp that reads the panels."""

        __props__ = ()
        # the 13 continuous CONSTRAINED parameters the data term depends on, in this order
        PARAMS = ("ab_n_perm", "ab_n_temp", "ab_n_rho", "ab_n_init", "ab_s_perm", "ab_s_rho", "ab_s_init",
                  "it_n_b", "it_n_d", "it_n_sigma", "it_s_b", "it_s_d", "it_s_sigma")

        def __init__(self, lazy: _Lazy):
            self.lazy = lazy

        def make_node(self, params, i_raw, waner):
            params = pt.as_tensor_variable(params)
            i_raw = pt.as_tensor_variable(i_raw)
            waner = pt.as_tensor_variable(waner)
            return Apply(self, [params, i_raw, waner], [pt.dscalar(), pt.dvector()])

        def perform(self, node, inputs, output_storage):
            params, i_raw, waner = inputs
            ll, g = self.lazy.loglik_dlogp_constrained(params, i_raw, waner)
            output_storage[0][0] = np.asarray(ll)
            output_storage[1][0] = g

        def grad(self, inputs, output_grads):
            params, i_raw, waner = inputs
            _, dparams = self(params, i_raw, waner)
            from pytensor.gradient import grad_undefined

            return [output_grads[0] * dparams, grad_undefined(self, 1, i_raw), grad_undefined(self, 2, waner)]

    def pymc_model(data, splits=None, ignore_pcrpos=False):
        """The reference's model with its likelihood evaluated on the MI355X."""
        from .data import check_splits

        check_splits(splits, data)
        op = AbdDataLogp(_Lazy(data, splits, ignore_pcrpos))
        with pm.Model(coords=data.coords) as model:
            p = pm.Beta("p", alpha=1, beta=data.n_gaps - 1)
            i_raw = pm.Bernoulli("i_raw", p, dims=("gap", "ind"))
            n_perm = pm.Gamma("ab_n_perm", mu=2.0, sigma=0.5)
            n_temp = pm.Gamma("ab_n_temp", mu=1.0, sigma=0.5)
            n_rho = pm.Beta("ab_n_rho", alpha=10.0, beta=1.0)
            n_init = pm.Normal("ab_n_init", -2, 1)
            s_perm = pm.Gamma("ab_s_perm", mu=2.0, sigma=0.5)
            s_rho = pm.Beta("ab_s_rho", alpha=10.0, beta=1.0)
            p_waner = pm.Beta("ab_s_p_waner", alpha=1.0, beta=1.0)
            waner = pm.Bernoulli("ab_s_waner", p=p_waner, dims="ind")
            pm.Gamma("ab_s_tempinf", mu=1.0, sigma=0.5)  # prior only (abd.py:272-274)
            pm.Gamma("ab_s_tempvac", mu=1.0, sigma=0.5)
            s_init = pm.Normal("ab_s_init", -2, 1)
            n_b, n_d, n_sig = pm.Normal("it_n_b", -1, 0.5), pm.Normal("it_n_d", 2, 0.5), pm.Exponential("it_n_sigma", 1)
            s_b, s_d, s_sig = pm.Normal("it_s_b", -1, 0.5), pm.Normal("it_s_d", 2, 0.5), pm.Exponential("it_s_sigma", 1)
            params = pt.stack([n_perm, n_temp, n_rho, n_init, s_perm, s_rho, s_init, n_b, n_d, n_sig, s_b, s_d, s_sig])
            ll, _ = op(params, i_raw, waner)
            pm.Potential("it_lik", ll)
        return model


def _theta_from_constrained(params, n_gaps):
    """13 constrained data-term parameters -> the 17-vector the C ABI takes (prior-only entries at their means)."""
    from .synthetic import theta_init

    t = theta_init(n_gaps)
    logit = lambda x: np.log(x / (1.0 - x))  # noqa: E731
    (n_perm, n_temp, n_rho, n_init, s_perm, s_rho, s_init, n_b, n_d, n_sig, s_b, s_d, s_sig) = [float(v) for v in params]
    t[1], t[2], t[3], t[4] = np.log(n_perm), np.log(n_temp), logit(n_rho), n_init
    t[5], t[6], t[10] = np.log(s_perm), logit(s_rho), s_init
    t[11], t[12], t[13] = n_b, n_d, np.log(n_sig)
    t[14], t[15], t[16] = s_b, s_d, np.log(s_sig)
    return t


def _grad_to_constrained(g_theta, params):
    """d/d theta (unconstrained) -> d/d constrained parameter for the 13 data-term entries."""
    (n_perm, n_temp, n_rho, n_init, s_perm, s_rho, s_init, n_b, n_d, n_sig, s_b, s_d, s_sig) = [float(v) for v in params]
    g = np.asarray(g_theta, dtype=float)
    return np.array([
        g[1] / n_perm, g[2] / n_temp, g[3] / (n_rho * (1 - n_rho)), g[4],
        g[5] / s_perm, g[6] / (s_rho * (1 - s_rho)), g[10],
        g[11], g[12], g[13] / n_sig, g[14], g[15], g[16] / s_sig,
    ])
