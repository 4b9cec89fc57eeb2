"""
A self-contained compound sampler for the model's two callables, used by ``abdpymc-infer`` when PyMC is not
installed: NUTS on the 17 continuous variables (calls ``logp_dlogp``) + binary Gibbs-Metropolis on
``[i_raw, ab_s_waner]`` (calls ``logp`` once per proposed flip) -- the step assignment ``pm.sample`` makes for
this model (reference call site abd.py:922).

The samplers only see callables, so they are tested on CPU against closed-form targets; on the GPU they are
driven by :class:`abdpymc_amd.model.AbdModel`.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------------
# NUTS (Hoffman & Gelman 2014, Algorithm 6: slice variant with dual-averaging step size), diagonal metric
# ---------------------------------------------------------------------------------------------------


class DualAveraging:
    def __init__(self, eps0: float, target: float = 0.8, gamma: float = 0.05, t0: float = 10.0, kappa: float = 0.75):
        self.mu = math.log(10.0 * eps0)
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.h_bar, self.log_eps_bar, self.m = 0.0, 0.0, 0
        self.eps = eps0

    def update(self, accept_stat: float) -> float:
        self.m += 1
        m = self.m
        self.h_bar = (1 - 1 / (m + self.t0)) * self.h_bar + (self.target - accept_stat) / (m + self.t0)
        log_eps = self.mu - math.sqrt(m) / self.gamma * self.h_bar
        eta = m ** (-self.kappa)
        self.log_eps_bar = eta * log_eps + (1 - eta) * self.log_eps_bar
        self.eps = math.exp(log_eps)
        return self.eps

    def final(self) -> float:
        return math.exp(self.log_eps_bar)


class Nuts:
    """One NUTS transition kernel over q in R^d for ``fn(q) -> (logp, grad)``."""

    def __init__(self, fn: Callable[[np.ndarray], Tuple[float, np.ndarray]], dim: int, rng: np.random.Generator,
                 max_treedepth: int = 10, target_accept: float = 0.8):
        self.fn, self.dim, self.rng = fn, dim, rng
        self.max_treedepth = max_treedepth
        self.inv_mass = np.ones(dim)  # diagonal M^-1 (posterior variances)
        self.da: Optional[DualAveraging] = None
        self.eps = 0.1
        self.target_accept = target_accept
        self.n_grad = 0

    def _grad(self, q):
        self.n_grad += 1
        lp, g = self.fn(q)
        if not np.isfinite(lp):
            return -np.inf, np.zeros_like(q)
        return float(lp), np.asarray(g, dtype=float)

    def _leapfrog(self, q, p, g, eps):
        p = p + 0.5 * eps * g
        q = q + eps * self.inv_mass * p
        lp, g = self._grad(q)
        p = p + 0.5 * eps * g
        return q, p, lp, g

    def _energy(self, lp, p):
        return lp - 0.5 * float(np.sum(self.inv_mass * p * p))

    def find_reasonable_eps(self, q, lp, g) -> float:
        eps = 0.1
        p = self.rng.standard_normal(self.dim) / np.sqrt(self.inv_mass)
        h0 = self._energy(lp, p)
        _, p1, lp1, _ = self._leapfrog(q, p, g, eps)
        dh = self._energy(lp1, p1) - h0
        a = 1.0 if (np.isfinite(dh) and dh > math.log(0.5)) else -1.0
        for _ in range(50):
            if not np.isfinite(dh):
                dh = -np.inf
            if a * dh <= -a * math.log(2.0):
                break
            eps *= 2.0**a
            _, p1, lp1, _ = self._leapfrog(q, p, g, eps)
            dh = self._energy(lp1, p1) - h0
        return eps

    def _build(self, q, p, g, log_u, v, j, eps, h0):
        if j == 0:
            q1, p1, lp1, g1 = self._leapfrog(q, p, g, v * eps)
            h1 = self._energy(lp1, p1)
            if not np.isfinite(h1):
                h1 = -np.inf
            n1 = int(log_u <= h1)
            s1 = log_u < h1 + 1000.0
            alpha = min(1.0, math.exp(min(0.0, h1 - h0))) if np.isfinite(h1) else 0.0
            return q1, p1, g1, q1, p1, g1, q1, lp1, g1, n1, s1, alpha, 1
        qm, pm, gm, qp, pp, gp, q1, lp1, g1, n1, s1, a1, na1 = self._build(q, p, g, log_u, v, j - 1, eps, h0)
        if s1:
            if v == -1:
                qm, pm, gm, _, _, _, q2, lp2, g2, n2, s2, a2, na2 = self._build(qm, pm, gm, log_u, v, j - 1, eps, h0)
            else:
                _, _, _, qp, pp, gp, q2, lp2, g2, n2, s2, a2, na2 = self._build(qp, pp, gp, log_u, v, j - 1, eps, h0)
            if n2 > 0 and self.rng.random() < n2 / max(n1 + n2, 1):
                q1, lp1, g1 = q2, lp2, g2
            a1, na1 = a1 + a2, na1 + na2
            dq = qp - qm
            s1 = s2 and float(np.dot(dq, self.inv_mass * pm)) >= 0 and float(np.dot(dq, self.inv_mass * pp)) >= 0
            n1 += n2
        return qm, pm, gm, qp, pp, gp, q1, lp1, g1, n1, s1, a1, na1

    def step(self, q, lp, g, adapt: bool):
        """One transition.  Returns (q, logp, grad, stats)."""
        eps = self.eps
        p0 = self.rng.standard_normal(self.dim) / np.sqrt(self.inv_mass)
        h0 = self._energy(lp, p0)
        log_u = h0 + math.log(self.rng.random() + 1e-300)
        qm = qp = q
        pm = pp = p0
        gm = gp = g
        j, n, s = 0, 1, True
        q_new, lp_new, g_new = q, lp, g
        alpha, n_alpha = 0.0, 1
        while s and j < self.max_treedepth:
            v = -1 if self.rng.random() < 0.5 else 1
            if v == -1:
                qm, pm, gm, _, _, _, q1, lp1, g1, n1, s1, alpha, n_alpha = self._build(qm, pm, gm, log_u, v, j, eps, h0)
            else:
                _, _, _, qp, pp, gp, q1, lp1, g1, n1, s1, alpha, n_alpha = self._build(qp, pp, gp, log_u, v, j, eps, h0)
            if s1 and self.rng.random() < min(1.0, n1 / n):
                q_new, lp_new, g_new = q1, lp1, g1
            n += n1
            dq = qp - qm
            s = s1 and float(np.dot(dq, self.inv_mass * pm)) >= 0 and float(np.dot(dq, self.inv_mass * pp)) >= 0
            j += 1
        acc = alpha / max(n_alpha, 1)
        if adapt and self.da is not None:
            self.eps = self.da.update(acc)
        return q_new, lp_new, g_new, dict(tree_depth=j, mean_tree_accept=acc, step_size=eps, n_steps=n_alpha,
                                          diverging=not s and j < self.max_treedepth and acc == 0.0)


# ---------------------------------------------------------------------------------------------------
# Binary Gibbs-Metropolis (PyMC's BinaryGibbsMetropolis.astep semantics: transit_p = 0.8, shuffled order)
# ---------------------------------------------------------------------------------------------------


def binary_gibbs_sweep(n_bits: int, flip_logp: Callable[[int], float], unflip: Callable[[int], None], logp_curr: float,
                       rng: np.random.Generator, transit_p: float = 0.8):
    """
    One sweep over all binary dims in random order.  ``flip_logp(idx)`` flips bit idx of the resident state
    and returns the joint logp there; ``unflip(idx)`` reverts a rejected flip.  Returns (logp, n_accepted,
    n_proposed).
    """
    order = rng.permutation(n_bits)
    propose = rng.random(n_bits) < transit_p
    n_acc = n_prop = 0
    for idx, do in zip(order, propose):
        if not do:
            continue  # same value proposed: nothing to evaluate
        n_prop += 1
        lp_prop = flip_logp(int(idx))
        d = lp_prop - logp_curr
        if np.isfinite(lp_prop) and (d >= 0 or math.log(rng.random() + 1e-300) < d):
            logp_curr = lp_prop
            n_acc += 1
        else:
            unflip(int(idx))
    return logp_curr, n_acc, n_prop


# ---------------------------------------------------------------------------------------------------
# Compound driver for one chain of the abd model
# ---------------------------------------------------------------------------------------------------


def sample_chain(model, chain: int, tune: int, draws: int, seed: int, record_deterministics: bool = True,
                 progress: Optional[Callable[[int, int], None]] = None, device_gibbs: bool = True) -> Dict[str, np.ndarray]:
    """Run tune + draws iterations of [NUTS; binary Gibbs] on chain slot ``chain`` of an AbdModel."""
    from .model import THETA_NAMES, constrain

    rng = np.random.default_rng([seed, chain])
    ctx = model.ctx
    G, N = model.n_gaps, model.n_inds
    pt = model.initial_point()
    i_raw = pt["i_raw"].astype(np.int8)
    waner = pt["ab_s_waner"].astype(np.int8)
    ctx.set_discrete(chain, i_raw, waner)
    q = model.ravel(pt) + rng.uniform(-1, 1, size=len(THETA_NAMES))  # start jitter U(-1, 1) on the value variables: pm.sample's default init 'jitter+adapt_diag'

    def fn(x):
        return ctx.logp_dlogp(chain, x)

    nuts = Nuts(fn, len(THETA_NAMES), rng)
    lp, g = fn(q)
    nuts.eps = nuts.find_reasonable_eps(q, lp, g)
    nuts.da = DualAveraging(nuts.eps)

    n_bits = G * N + N
    use_device_sweep = hasattr(ctx, "gibbs_sweep") and device_gibbs

    # host-driven fallback (one joint-logp evaluation per proposed flip, the reference's cost model)
    def flip_logp(idx):
        ctx.flip_discrete(chain, idx)
        if idx < G * N:
            i_raw.ravel()[idx] ^= 1
        else:
            waner[idx - G * N] ^= 1
        return ctx.logp(chain, q)

    def unflip(idx):
        ctx.flip_discrete(chain, idx)
        if idx < G * N:
            i_raw.ravel()[idx] ^= 1
        else:
            waner[idx - G * N] ^= 1

    out_q = np.empty((draws, len(THETA_NAMES)))
    out_i_raw = np.empty((draws, G, N), dtype=np.int8)
    out_w = np.empty((draws, N), dtype=np.int8)
    stats = {k: np.empty(draws) for k in ("lp", "tree_depth", "mean_tree_accept", "step_size", "n_steps", "gibbs_accept")}
    det = None
    if record_deterministics:
        det = dict(i=np.empty((draws, G, N), dtype=np.int8), ab_n_mu=np.empty((draws, G, N)), ab_s_mu=np.empty((draws, G, N)))

    # mass-matrix adaptation: variances of the tuning draws in expanding windows
    window_start, window_len = max(10, tune // 10), max(20, tune // 8)
    window = []
    for it in range(tune + draws):
        tuning = it < tune
        q, lp, g, st = nuts.step(q, lp, g, adapt=tuning)
        if use_device_sweep:
            # the whole sweep in one launch: each individual's G + 1 proposals on its own wave (abd_gibbs.hpp)
            acc, prop = ctx.gibbs_sweep([chain], q[None, :], seed=(seed << 20) ^ 0x5EED, sweep=it)
            n_acc, n_prop = int(acc[0]), int(prop[0])
            if not tuning:
                i_raw, waner = ctx.get_discrete(chain)
        else:
            lp, n_acc, n_prop = binary_gibbs_sweep(n_bits, flip_logp, unflip, lp, rng)
        lp, g = fn(q)  # logp and gradient at the new discrete state
        if tuning:
            if it >= window_start:
                window.append(q.copy())
            if len(window) >= window_len and it < tune - 20:
                var = np.var(np.asarray(window), axis=0)
                nuts.inv_mass = (len(window) / (len(window) + 5.0)) * var + 1e-3 * (5.0 / (len(window) + 5.0))
                window, window_len = [], int(window_len * 1.5)
                nuts.eps = nuts.find_reasonable_eps(q, lp, g)
                nuts.da = DualAveraging(nuts.eps)
            if it == tune - 1:
                nuts.eps = nuts.da.final()
        else:
            k = it - tune
            out_q[k] = q
            out_i_raw[k] = i_raw
            out_w[k] = waner
            stats["lp"][k] = lp
            for name in ("tree_depth", "mean_tree_accept", "step_size", "n_steps"):
                stats[name][k] = st[name]
            stats["gibbs_accept"][k] = n_acc / max(n_prop, 1)
            if det is not None:
                d_i, d_n, d_s = ctx.deterministics(chain, q)
                det["i"][k], det["ab_n_mu"][k], det["ab_s_mu"][k] = d_i, d_n, d_s
        if progress is not None:
            progress(it + 1, tune + draws)

    res = {name: out_q[:, k].copy() for k, name in enumerate(THETA_NAMES)}
    res.update(constrain(out_q))
    res["i_raw"] = out_i_raw
    res["ab_s_waner"] = out_w
    if det is not None:
        res.update(det)
    res.update({f"stat_{k}": v for k, v in stats.items()})
    res["n_grad_evals"] = np.asarray(nuts.n_grad)
    return res


def record_bytes(chains: int, n_rec: int, G: int, N: int, record_deterministics: bool, record_discrete: bool) -> int:
    """Host bytes of the per-draw arrays of a run that records ``n_rec`` draws per chain: i (int8) + ab_n_mu + ab_s_mu
    (float64) per cell for the Deterministics, i_raw (int8) per cell + ab_s_waner (int8) per individual for the discrete state."""
    per_draw = (G * N * 17 if record_deterministics else 0) + (G * N + N if record_discrete else 0)
    return chains * n_rec * per_draw


def record_budget_bytes() -> int:
    """What the per-draw arrays of one process may take on the host (``ABD_RECORD_BUDGET_GB``, default 8 GiB)."""
    import os

    return int(float(os.environ.get("ABD_RECORD_BUDGET_GB", "8")) * 2 ** 30)


def sample_native(model, tune: int, draws: int, chains: int = 1, seed: int = 0, record_deterministics: bool = True,
                  record_discrete: bool = True, progress: Optional[Callable[[int, int, int], None]] = None,
                  target_accept: float = 0.8, max_treedepth: int = 10, chunk: int = 50,
                  chain_offset: int = 0, dense_metric: bool = False, thin: int = 1,
                  budget_bytes: Optional[int] = None) -> Dict[str, np.ndarray]:
    """
    The compound step inside the library (``abd_sampler_*``): the chains advance as independent units, each at its own
    pace (NUTS transitions as leapfrog trains on the device, the Gibbs sweep, the re-evaluation at the new state); nothing
    a chain draws depends on the others.  Posterior means of the three Deterministics over ALL draws are accumulated on
    the device and returned as ``mean_i``, ``mean_ab_n_mu``, ``mean_ab_s_mu`` (what the reference's downstream analysis
    reads: survival.py:64-69, 105-114); per-draw copies of the (gap, ind) arrays only when asked for, and only of every
    ``thin``-th draw (draws 0, thin, 2 thin, ...: ``draw_index``) -- the reference keeps every draw and thins afterwards
    (subsample_idata.py); at BASELINE config 3 a draw is 36 MB per chain, so here it is done while sampling.  The 17
    value variables and the sampler statistics are kept for every draw.  A run whose per-draw arrays would exceed
    ``budget_bytes`` on the host (default ``record_budget_bytes()``) is refused with the ``thin`` that would fit.
    ``chain_offset`` is the global id of this process's first chain when chains are sharded over GPUs: local
    chain c uses the random streams of global chain ``chain_offset + c``.
    """
    from .model import THETA_NAMES, constrain

    ctx = model.ctx
    G, N = model.n_gaps, model.n_inds
    if thin < 1:
        raise ValueError(f"thin must be >= 1, got {thin}")
    n_rec = (draws + thin - 1) // thin
    if record_deterministics or record_discrete:
        budget = record_budget_bytes() if budget_bytes is None else int(budget_bytes)
        need = record_bytes(chains, n_rec, G, N, record_deterministics, record_discrete)
        if need > budget:
            per_draw = need // max(n_rec, 1)
            fit = max(1, budget // max(per_draw, 1))          # draws per chain set that fit
            thin_fit = -(-draws // fit)
            raise ValueError(
                f"recording {n_rec} draws of {chains} chains x ({G}, {N}) needs {need / 2 ** 30:.1f} GiB of host arrays, over the "
                f"budget of {budget / 2 ** 30:.1f} GiB: thin >= {thin_fit} fits (or record less: no_deterministics / "
                f"record_discrete=False; the posterior means are returned either way; ABD_RECORD_BUDGET_GB raises the budget)")
    chunk = max(thin, chunk - chunk % thin) if thin > 1 else chunk  # calls record iterations 0, thin, ... of THEIR range
    pt = model.initial_point()
    q0 = np.empty((chains, len(THETA_NAMES)))
    for c in range(chains):
        rng = np.random.default_rng([seed, chain_offset + c])
        ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
        q0[c] = model.ravel(pt) + rng.uniform(-1, 1, size=len(THETA_NAMES))  # start jitter U(-1, 1) on the value variables: pm.sample's default init 'jitter+adapt_diag'
    smp = ctx.sampler(np.arange(chains), q0, tune=tune, seed=seed, target_accept=target_accept,
                      max_treedepth=max_treedepth, gibbs=True, accumulate=True, chain_offset=chain_offset,
                      dense_metric=dense_metric)
    n_grad = chains  # the evaluation at the starting points
    done = 0

    def advance(n):
        nonlocal done, n_grad
        th, st = smp.run(n)
        n_grad += int(st["n_steps"].sum()) + n * chains  # leapfrogs + the re-evaluation after each sweep
        done += n
        if progress is not None:
            for c in range(chains):
                progress(c, done, tune + draws)
        return th, st

    left = tune
    while left > 0:
        advance(min(chunk, left))
        left -= min(chunk, left)
    thetas, stats = [], []
    out_i_raw = np.empty((chains, n_rec, G, N), dtype=np.int8) if record_discrete else None
    out_w = np.empty((chains, n_rec, N), dtype=np.int8) if record_discrete else None
    det = None
    if record_deterministics:
        det = dict(i=np.empty((chains, n_rec, G, N), dtype=np.int8), ab_n_mu=np.empty((chains, n_rec, G, N)),
                   ab_s_mu=np.empty((chains, n_rec, G, N)))
    k = 0
    while k < draws:
        n = min(chunk, draws - k)
        if record_deterministics or record_discrete:
            # staged on the device, copied out in large blocks straight into the arrays above
            th, st = smp.run_record(n, k // thin, i_raw=out_i_raw, ab_s_waner=out_w, thin=thin, **(det or {}))
            n_grad += int(st["n_steps"].sum()) + n * chains
            done += n
            if progress is not None:
                for c in range(chains):
                    progress(c, done, tune + draws)
        else:
            th, st = advance(n)
        thetas.append(th)
        stats.append(st)
        k += n
    out_q = np.concatenate(thetas, axis=1) if thetas else np.empty((chains, 0, len(THETA_NAMES)))
    res = {name: out_q[:, :, j].copy() for j, name in enumerate(THETA_NAMES)}
    res.update(constrain(out_q))
    if record_discrete:
        res["i_raw"], res["ab_s_waner"] = out_i_raw, out_w
    if det is not None:
        res.update(det)
    if record_discrete or det is not None:
        res["draw_index"] = np.tile(np.arange(0, draws, thin, dtype=np.int64), (chains, 1))  # which draws the (gap, ind) arrays hold
    if draws:
        means = [smp.means(c) for c in range(chains)]
        for j, name in enumerate(("mean_i", "mean_ab_n_mu", "mean_ab_s_mu")):
            res[name] = np.stack([m[j] for m in means])
        cat = {name: np.concatenate([s[name] for s in stats], axis=1) for name in stats[0]}
        for name in ("lp", "tree_depth", "mean_tree_accept", "step_size", "n_steps", "diverging", "energy"):
            res[f"stat_{name}"] = cat[name]
        res["stat_gibbs_accept"] = cat["gibbs_accepted"] / np.maximum(cat["gibbs_proposed"], 1)
    res["n_grad_evals"] = np.full(chains, n_grad // chains)
    smp.close()
    return res


def sample(model, tune: int, draws: int, chains: int = 1, seed: int = 0, record_deterministics: bool = True,
           progress: Optional[Callable[[int, int, int], None]] = None, device_gibbs: bool = True,
           native: bool = True, record_discrete: bool = True, chain_offset: int = 0,
           dense_metric: bool = False, thin: int = 1, budget_bytes: Optional[int] = None) -> Dict[str, np.ndarray]:
    """``pm.sample(tune, draws)`` for the abd model: returns arrays with leading (chain, draw) axes (the per-draw
    (gap, ind) arrays hold every ``thin``-th draw: ``sample_native``)."""
    if chains > model.n_chains:
        raise ValueError(f"model was built with {model.n_chains} chain slots, {chains} requested")
    if native and device_gibbs and hasattr(model.ctx, "sampler"):
        return sample_native(model, tune, draws, chains, seed, record_deterministics, record_discrete, progress,
                             chain_offset=chain_offset, dense_metric=dense_metric, thin=thin, budget_bytes=budget_bytes)
    if chain_offset or dense_metric or thin != 1:
        raise ValueError("chain_offset / dense_metric / thin need the native sampler")
    per_chain = []
    for c in range(chains):
        cb = (lambda a, b, c=c: progress(c, a, b)) if progress else None
        per_chain.append(sample_chain(model, c, tune, draws, seed, record_deterministics, cb, device_gibbs))
    return {k: np.stack([pc[k] for pc in per_chain]) for k in per_chain[0]}
