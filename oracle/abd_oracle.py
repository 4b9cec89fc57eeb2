"""
CPU oracle for the abdpymc joint log-probability hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the algorithm in the reference's ``abdpymc/abd.py``
(davipatti/abdpymc @ 2025-08-24).  It exists so that the HIP path can be checked against an
independent CPU implementation.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``abdpymc_amd`` never does.

Pinning status
--------------
* Integer pre-pass and temp/perm responses: PINNED by the reference's own known-answer tests
  (``abdpymc/test_abd.py:17-63, 103-146, 199-206, 278-366, 370-413, 444-617, 770-1040``), restated
  as data in ``tests/golden/reference_known_answers.json`` and checked by ``tests/test_oracle_golden.py``.
* ``logistic``/``invlogistic``: PINNED by ``test_abd.py:625-630``.
* Joint logp / dlogp: **parity unpinned** -- no reference test evaluates ``model.logp`` and PyMC /
  PyTensor are not installed here.  The densities follow PyMC v5's published closed forms
  (``pymc`` is an unpinned dependency, ``pyproject.toml:10``) and are cross-checked against
  ``scipy.stats`` and central finite differences in ``tests/test_oracle_logp.py``.

Two forms are provided:

* *faithful*: literal restatement of the dense ``(G, G, N)`` design used by the reference model
  (``abd.py:224-274``), including the quirk that ``_temp_response_vector_rho`` ignores ``temp``.
* *recurrence*: the O(G*N) scan ``T[g] = rho*T[g-1] + temp*e[g]`` (``abd.py:277-293``) with forward
  sensitivities, used for the analytic gradient and for sizes where the dense form does not fit.

Conventions: G = n_gaps, N = n_inds, arrays shaped (G, N) are gap-major exactly as in the reference.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Optional, Sequence

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)

# Order of the 17 continuous value variables (PyMC v5 naming, creation order in abd.py:424-467).
THETA_NAMES = (
    "p_logodds__",  # 0   abd.py:424
    "ab_n_perm_log__",  # 1   abd.py:329
    "ab_n_temp_log__",  # 2   abd.py:333
    "ab_n_rho_logodds__",  # 3   abd.py:334
    "ab_n_init",  # 4   abd.py:340
    "ab_s_perm_log__",  # 5   abd.py:367
    "ab_s_rho_logodds__",  # 6   abd.py:371
    "ab_s_p_waner_logodds__",  # 7   abd.py:372
    "ab_s_tempinf_log__",  # 8   abd.py:377
    "ab_s_tempvac_log__",  # 9   abd.py:383
    "ab_s_init",  # 10  abd.py:388
    "it_n_b",  # 11  abd.py:464
    "it_n_d",  # 12  abd.py:465
    "it_n_sigma_log__",  # 13  abd.py:467
    "it_s_b",  # 14
    "it_s_d",  # 15
    "it_s_sigma_log__",  # 16
)
N_THETA = len(THETA_NAMES)


# --------------------------------------------------------------------------------------------
# Containers
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass
class AntigenObs:
    """Observation list of one antigen (abd.py:22-43): od readings gathered out of (G, N)."""

    idx_gap: np.ndarray  # (K,) int   = df.elapsed_months   abd.py:35
    idx_ind: np.ndarray  # (K,) int   = df.individual_i     abd.py:36
    log_dilution: np.ndarray  # (K,) float                    abd.py:462
    od: np.ndarray  # (K,) float                    abd.py:468


@dataclasses.dataclass
class Cohort:
    """What abd.model() consumes (abd.py:396-442)."""

    n_gaps: int
    n_inds: int
    vacs: np.ndarray  # (N, G) 0/1   abd.py:114
    pcrpos: np.ndarray  # (N, G) 0/1   abd.py:115
    s: AntigenObs
    n: AntigenObs


# --------------------------------------------------------------------------------------------
# Tensor building blocks (abd.py:224-306, 552-557)
# --------------------------------------------------------------------------------------------


def make_decay_design(n_gaps: int) -> np.ndarray:
    """abd.py:224-239: D[r, c] = max(0, c - r)."""
    a = np.arange(n_gaps)
    return np.maximum(0, a - a[:, None])


def temp_response_scalar_rho(exposure, temp, rho) -> np.ndarray:
    """abd.py:242-260 (dense form, scalar rho)."""
    exposure = np.asarray(exposure, dtype=float)
    design = make_decay_design(exposure.shape[0])
    offset = np.tril(np.ones_like(design), -1)
    responses_each_gap = (rho**design - offset) * temp
    return (responses_each_gap[:, :, None] * exposure[:, None, :]).sum(axis=0)


def temp_response_vector_rho(exposure, temp, rho) -> np.ndarray:
    """abd.py:263-274 (dense form, per-individual rho).  ``temp`` is accepted and IGNORED,
    exactly as in the reference (the expression at abd.py:272 never multiplies by it)."""
    exposure = np.asarray(exposure, dtype=float)
    rho = np.asarray(rho, dtype=float)
    design = make_decay_design(exposure.shape[0])
    offset = np.tril(np.ones_like(design), -1)
    return (
        (rho ** design[..., None] - offset[..., None]) * exposure[:, None, :]
    ).sum(axis=0)


def temp_response_scan(exposure, temp, rho) -> np.ndarray:
    """abd.py:277-293: prev * rho + e * temp, initial state zeros(n_inds).  rho may be (N,)."""
    exposure = np.asarray(exposure, dtype=float)
    out = np.empty_like(exposure)
    prev = np.zeros(exposure.shape[1])
    for g in range(exposure.shape[0]):
        prev = prev * rho + exposure[g] * temp
        out[g] = prev
    return out


def perm_response(exposure, perm) -> np.ndarray:
    """abd.py:296-306."""
    return np.where(np.cumsum(exposure, axis=0) > 0.0, perm, 0.0)


def logistic(x, a, b, d):
    """abd.py:556-557."""
    return d / (1 + np.exp(-b * (x - a)))


def invlogistic(x, a, b, d):
    """abd.py:552-553."""
    return a - np.log(d / x - 1) / b


# --------------------------------------------------------------------------------------------
# Infection constraints (abd.py:560-882)
# --------------------------------------------------------------------------------------------


def mask_future_infection(i0, im3, im2, im1):
    """abd.py:581-601: switch(im3 | im2 | im1, 0, i0) -- passes non-binary i0 through."""
    return np.where(np.asarray(im3) | np.asarray(im2) | np.asarray(im1), 0, i0)


def mask_three_gaps(arr) -> np.ndarray:
    """abd.py:560-578: scan with taps -3,-2,-1 on its OWN OUTPUT, initial three zero rows, int8."""
    arr = np.asarray(arr).astype(np.int8)
    n_gaps, n_inds = arr.shape
    hist = np.zeros((3, n_inds), dtype=np.int8)  # rows: t-3, t-2, t-1
    out = np.empty_like(arr)
    for t in range(n_gaps):
        cur = mask_future_infection(arr[t], hist[0], hist[1], hist[2]).astype(np.int8)
        out[t] = cur
        hist[0], hist[1], hist[2] = hist[1].copy(), hist[2].copy(), cur
    return out


def mask_multiple_infections(arr) -> np.ndarray:
    """abd.py:792-818."""
    arr = np.asarray(arr)
    if arr.shape[0] == 0:
        return arr
    return np.where(arr.cumsum(axis=0) > 1, 0, arr)


def mask_multiple_infections_2_chunks(arr, split) -> np.ndarray:
    """abd.py:821-862."""
    arr = np.asarray(arr)
    return np.concatenate(
        (mask_multiple_infections(arr[:split]), mask_multiple_infections(arr[split:]))
    )


def mask_multiple_infections_3_chunks(arr, split0, split1) -> np.ndarray:
    """abd.py:774-789."""
    arr = np.asarray(arr)
    return np.concatenate(
        (
            mask_multiple_infections(arr[:split0]),
            mask_multiple_infections(arr[split0:split1]),
            mask_multiple_infections(arr[split1:]),
        )
    )


def incorporate_pcrpos(i_raw, pcrpos) -> np.ndarray:
    """abd.py:732-771: where(pcrpos.any(axis=0), pcrpos, i_raw)."""
    i_raw = np.asarray(i_raw)
    pcrpos = np.asarray(pcrpos)
    return np.where(pcrpos.any(axis=0), pcrpos, i_raw)


def check_splits(splits, n_gaps: Optional[int] = None) -> None:
    """abd.py:604-622 (same conditions, same messages)."""
    if splits is not None:
        if any(split < 0 for split in splits):
            raise ValueError("split indexes must be positive")
        if sorted(splits) != list(splits):
            raise ValueError("splits must be in ascending order")
        if n_gaps is not None and splits and splits[-1] > n_gaps:
            raise ValueError(
                f"largest split must be less than n_gaps - 1, ({splits[-1]})"
            )
        if len(splits) != len(set(splits)):
            raise ValueError("splits not unique")
        if any(not isinstance(split, int) for split in splits):
            raise ValueError("splits must be ints")


def chunk_bounds(splits, n_gaps):
    """Chunk [lo, hi) bounds for 0-2 splits (abd.py:685-686, 713-715)."""
    edges = [0, *list(splits or ()), n_gaps]
    return [(edges[k], edges[k + 1]) for k in range(len(edges) - 1)]


def constrain_infections(i_raw, pcrpos_gn, splits=None) -> np.ndarray:
    """
    OneTimeChunk.constrain_infections (abd.py:640-649) for no splits,
    MultipleTimeChunks.constrain_infections (abd.py:658-667) with Two/ThreeTimeChunks
    (abd.py:670-729) otherwise.  ``pcrpos_gn`` is (G, N), i.e. ``data.pcrpos.T`` (abd.py:416-418).
    Returns int8 (G, N): the Deterministic "i".
    """
    i_raw = np.asarray(i_raw)
    pcrpos_gn = np.asarray(pcrpos_gn)
    if splits is None or len(splits) == 0:
        i_pcrpos = i_raw + pcrpos_gn
        i0 = np.where(i_pcrpos > 0.0, 1.0, 0.0)
        return mask_three_gaps(i0)
    if len(splits) > 2:
        raise NotImplementedError("only implemented 1-3 time chunks (0-2 splits)")
    parts = []
    for lo, hi in chunk_bounds(splits, i_raw.shape[0]):
        chunk = mask_multiple_infections(i_raw[lo:hi])
        parts.append(incorporate_pcrpos(chunk, pcrpos_gn[lo:hi]))
    return mask_three_gaps(np.concatenate(parts))


# --------------------------------------------------------------------------------------------
# Parameter transforms and priors (PyMC v5 closed forms; Jacobians included as pm.sample uses)
# --------------------------------------------------------------------------------------------


def _sigmoid(t):
    return 1.0 / (1.0 + math.exp(-t))


def constrained(theta) -> dict:
    """Backward transforms of T1 (logodds -> sigmoid, log -> exp)."""
    t = [float(v) for v in theta]
    return dict(
        p=_sigmoid(t[0]),
        perm_n=math.exp(t[1]),
        temp_n=math.exp(t[2]),
        rho_n=_sigmoid(t[3]),
        init_n=t[4],
        perm_s=math.exp(t[5]),
        rho_s=_sigmoid(t[6]),
        p_waner=_sigmoid(t[7]),
        tempinf=math.exp(t[8]),
        tempvac=math.exp(t[9]),
        init_s=t[10],
        b_n=t[11],
        d_n=t[12],
        sigma_n=math.exp(t[13]),
        b_s=t[14],
        d_s=t[15],
        sigma_s=math.exp(t[16]),
    )


def _gamma_ab(mu, sigma):
    """PyMC Gamma(mu, sigma) -> (alpha, beta=rate)."""
    return mu * mu / (sigma * sigma), mu / (sigma * sigma)


def _lbeta(a, b):
    return math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b)


def prior_logp_literal(theta, n_gaps, n_cells, n1, n_inds, m1) -> float:
    """
    Sum of every prior term + transform log-Jacobian, written the way PyMC evaluates them
    (sigmoid then log / log1p), i.e. the *literal* form.  Terms: abd.py:424, 427, 329-340, 367-388,
    464-467.  ``n1`` = sum(i_raw), ``m1`` = sum(waner), ``n_cells`` = G*N.
    """
    c = constrained(theta)
    t = [float(v) for v in theta]
    lp = 0.0

    def beta_logodds(x, a, b):
        # Beta logp (switch on a==1 / b==1 as PyMC does) + logodds log|J| = log x + log1p(-x)
        v = 0.0 if a == 1.0 else (a - 1.0) * math.log(x)
        v += 0.0 if b == 1.0 else (b - 1.0) * math.log1p(-x)
        v -= _lbeta(a, b)
        return v + math.log(x) + math.log1p(-x)

    def gamma_log(th, mu, sigma):
        a, b = _gamma_ab(mu, sigma)
        x = math.exp(th)
        return -math.lgamma(a) + a * math.log(b) - b * x + (a - 1.0) * math.log(x) + th

    def normal(x, mu, sigma):
        return -0.5 * ((x - mu) / sigma) ** 2 - math.log(math.sqrt(2.0 * math.pi)) - math.log(sigma)

    # p ~ Beta(1, G-1), i_raw ~ Bernoulli(p) on the RAW matrix          abd.py:424-427
    lp += beta_logodds(c["p"], 1.0, float(n_gaps - 1))
    lp += n1 * math.log(c["p"]) + (n_cells - n1) * math.log1p(-c["p"])
    # N response priors                                                    abd.py:329-340
    lp += gamma_log(t[1], 2.0, 0.5)
    lp += gamma_log(t[2], 1.0, 0.5)
    lp += beta_logodds(c["rho_n"], 10.0, 1.0)
    lp += normal(t[4], -2.0, 1.0)
    # S response priors                                                    abd.py:367-388
    lp += gamma_log(t[5], 2.0, 0.5)
    lp += beta_logodds(c["rho_s"], 10.0, 1.0)
    lp += beta_logodds(c["p_waner"], 1.0, 1.0)
    lp += m1 * math.log(c["p_waner"]) + (n_inds - m1) * math.log1p(-c["p_waner"])
    lp += gamma_log(t[8], 1.0, 0.5)
    lp += gamma_log(t[9], 1.0, 0.5)
    lp += normal(t[10], -2.0, 1.0)
    # sigmoid priors, n then s                                             abd.py:464-467
    for o in (11, 14):
        lp += normal(t[o], -1.0, 0.5)
        lp += normal(t[o + 1], 2.0, 0.5)
        lp += -math.exp(t[o + 2]) + t[o + 2]  # Exponential(1) + log|J|
    return lp


def _softplus(t):
    return max(t, 0.0) + math.log1p(math.exp(-abs(t)))


def prior_logp_grad(theta, n_gaps, n_cells, n1, n_inds, m1):
    """Closed-form priors with log sigmoid written as -softplus, and their gradient."""
    t = [float(v) for v in theta]
    g = np.zeros(N_THETA)
    lp = 0.0

    def l0l1(th):  # log sigmoid(th), log(1 - sigmoid(th)), sigmoid(th)
        return -_softplus(-th), -_softplus(th), _sigmoid(th)

    # theta0
    L0, L1, p = l0l1(t[0])
    bm1 = float(n_gaps - 1) - 1.0
    lp += bm1 * L1 - _lbeta(1.0, float(n_gaps - 1)) + L0 + L1 + n1 * L0 + (n_cells - n1) * L1
    g[0] = (1.0 + n1) * (1.0 - p) - p * (bm1 + 1.0 + (n_cells - n1))
    # Gammas
    for k, (mu, sd) in ((1, (2.0, 0.5)), (2, (1.0, 0.5)), (5, (2.0, 0.5)), (8, (1.0, 0.5)), (9, (1.0, 0.5))):
        a, b = _gamma_ab(mu, sd)
        x = math.exp(t[k])
        lp += a * math.log(b) - math.lgamma(a) + a * t[k] - b * x
        g[k] = a - b * x
    # Beta(10,1) rho
    for k in (3, 6):
        L0, L1, r = l0l1(t[k])
        lp += 9.0 * L0 - _lbeta(10.0, 1.0) + L0 + L1
        g[k] = 10.0 * (1.0 - r) - r
    # p_waner
    L0, L1, q = l0l1(t[7])
    lp += L0 + L1 + m1 * L0 + (n_inds - m1) * L1
    g[7] = (1.0 + m1) * (1.0 - q) - q * (1.0 + (n_inds - m1))
    # Normals
    for k, mu, sd in ((4, -2.0, 1.0), (10, -2.0, 1.0), (11, -1.0, 0.5), (12, 2.0, 0.5), (14, -1.0, 0.5), (15, 2.0, 0.5)):
        z = (t[k] - mu) / sd
        lp += -0.5 * z * z - math.log(sd) - 0.5 * LOG_2PI
        g[k] = -z / sd
    # Exponential(1) on sigma, log transform
    for k in (13, 16):
        x = math.exp(t[k])
        lp += -x + t[k]
        g[k] = -x + 1.0
    return lp, g


# --------------------------------------------------------------------------------------------
# Joint logp
# --------------------------------------------------------------------------------------------


def _normal_lik(od, mu, sigma):
    r = (od - mu) / sigma
    return float(np.sum(-0.5 * r * r) - od.size * (math.log(sigma) + 0.5 * LOG_2PI))


def deterministics(theta, i_raw, waner, cohort: Cohort, splits=None, ignore_pcrpos=False, dense=False):
    """
    The three recorded Deterministics (abd.py:649/667, 341, 389-391): i (G,N) int8,
    ab_n_mu (G,N), ab_s_mu (G,N).  ``dense=True`` uses the literal (G,G,N) forms.
    """
    c = constrained(theta)
    v = np.asarray(cohort.vacs, dtype=float).T  # abd.py:413
    pcr = np.zeros_like(v) if ignore_pcrpos else np.asarray(cohort.pcrpos, dtype=float).T  # abd.py:416-418
    i = constrain_infections(np.asarray(i_raw), pcr, splits)
    w = np.asarray(waner, dtype=float)
    rho_per_ind = c["rho_s"] * w + 1 - w  # abd.py:374
    fi = i.astype(float)
    if dense:
        tn = temp_response_scalar_rho(fi, c["temp_n"], c["rho_n"])
        ti = temp_response_vector_rho(fi, c["tempinf"], rho_per_ind)
        tv = temp_response_vector_rho(v, c["tempvac"], rho_per_ind)
    else:
        tn = temp_response_scan(fi, c["temp_n"], c["rho_n"])
        ti = temp_response_scan(fi, 1.0, rho_per_ind)  # temp ignored (abd.py:272)
        tv = temp_response_scan(v, 1.0, rho_per_ind)
    mu_n = perm_response(fi, c["perm_n"]) + tn + c["init_n"]  # abd.py:341
    mu_s = perm_response(fi + v, c["perm_s"]) + ti + tv + c["init_s"]  # abd.py:389-391
    return i, mu_n, mu_s


def joint_logp(theta, i_raw, waner, cohort: Cohort, splits=None, ignore_pcrpos=False, dense=True) -> float:
    """Scalar joint logp as Model.compile_logp() would return it, literal form."""
    c = constrained(theta)
    i_raw = np.asarray(i_raw)
    waner = np.asarray(waner)
    _, mu_n, mu_s = deterministics(theta, i_raw, waner, cohort, splits, ignore_pcrpos, dense=dense)
    lp = prior_logp_literal(
        theta, cohort.n_gaps, i_raw.size, int(i_raw.sum()), waner.size, int(waner.sum())
    )
    for obs, mu, b, d, sig in (
        (cohort.n, mu_n, c["b_n"], c["d_n"], c["sigma_n"]),
        (cohort.s, mu_s, c["b_s"], c["d_s"], c["sigma_s"]),
    ):
        a = mu[obs.idx_gap, obs.idx_ind]  # abd.py:343, 393
        m = logistic(obs.log_dilution, a, b, d)  # abd.py:461-466
        lp += _normal_lik(obs.od, m, sig)
    return float(lp)


def _scan_with_sens(e, rho_vec):
    """U[g] = rho*U[g-1] + e[g] and D[g] = dU[g]/drho = rho*D[g-1] + U[g-1] (per column)."""
    G, N = e.shape
    U = np.empty((G, N))
    D = np.empty((G, N))
    u = np.zeros(N)
    dd = np.zeros(N)
    for g in range(G):
        dd = rho_vec * dd + u
        u = rho_vec * u + e[g]
        U[g] = u
        D[g] = dd
    return U, D


def logp_dlogp(theta, i_raw, waner, cohort: Cohort, splits=None, ignore_pcrpos=False):
    """
    (logp, grad[17]) as Model.logp_dlogp_function() would return them: recurrence
    form + analytic gradient with forward sensitivities.
    """
    c = constrained(theta)
    t = [float(x) for x in theta]
    i_raw = np.asarray(i_raw)
    waner = np.asarray(waner)
    G, N = cohort.n_gaps, cohort.n_inds
    v = np.asarray(cohort.vacs, dtype=float).T
    pcr = np.zeros_like(v) if ignore_pcrpos else np.asarray(cohort.pcrpos, dtype=float).T
    i = constrain_infections(i_raw, pcr, splits).astype(float)
    w = waner.astype(float)

    lp, grad = prior_logp_grad(t, G, i_raw.size, int(i_raw.sum()), waner.size, int(waner.sum()))

    # N: T = temp_n * U_n ; scalar rho
    Un, Dn = _scan_with_sens(i, np.full(N, c["rho_n"]))
    cum_n = (np.cumsum(i, axis=0) > 0).astype(float)
    mu_n = cum_n * c["perm_n"] + c["temp_n"] * Un + c["init_n"]
    # S: unit boosts (temp ignored), rho_j = rho_s if waner else 1
    rho_j = c["rho_s"] * w + 1 - w
    Us, Ds = _scan_with_sens(i + v, rho_j)
    Ds = Ds * w  # d rho_j / d rho_s = w
    cum_s = (np.cumsum(i + v, axis=0) > 0).astype(float)
    mu_s = cum_s * c["perm_s"] + Us + c["init_s"]

    def lik(obs, mu, b, d, sig):
        a = mu[obs.idx_gap, obs.idx_ind]
        x, y = obs.log_dilution, obs.od
        e = np.exp(b * (a - x))
        s = 1.0 / (1.0 + e)
        m = d * s
        r = (y - m) / sig
        ll = float(np.sum(-0.5 * r * r) - y.size * (math.log(sig) + 0.5 * LOG_2PI))
        wgt = r / sig  # d ll / d m
        dm_du = -d * s * (e * s)  # 1 - s = e*s
        ga = wgt * dm_du * b
        gb = float(np.sum(wgt * dm_du * (a - x)))
        gd = float(np.sum(wgt * s))
        gls = float(np.sum(r * r - 1.0))
        return ll, ga, gb, gd, gls

    ll, ga, gb, gd, gls = lik(cohort.n, mu_n, c["b_n"], c["d_n"], c["sigma_n"])
    lp += ll
    kg, ki = cohort.n.idx_gap, cohort.n.idx_ind
    grad[1] += float(np.sum(ga * cum_n[kg, ki])) * c["perm_n"]
    grad[2] += float(np.sum(ga * Un[kg, ki])) * c["temp_n"]
    grad[3] += float(np.sum(ga * Dn[kg, ki])) * c["temp_n"] * c["rho_n"] * (1 - c["rho_n"])
    grad[4] += float(np.sum(ga))
    grad[11] += gb
    grad[12] += gd
    grad[13] += gls

    ll, ga, gb, gd, gls = lik(cohort.s, mu_s, c["b_s"], c["d_s"], c["sigma_s"])
    lp += ll
    kg, ki = cohort.s.idx_gap, cohort.s.idx_ind
    grad[5] += float(np.sum(ga * cum_s[kg, ki])) * c["perm_s"]
    grad[6] += float(np.sum(ga * Ds[kg, ki])) * c["rho_s"] * (1 - c["rho_s"])
    grad[10] += float(np.sum(ga))
    grad[14] += gb
    grad[15] += gd
    grad[16] += gls
    return float(lp), grad


def finite_difference_grad(theta, i_raw, waner, cohort, splits=None, ignore_pcrpos=False, h=1e-5, dense=False):
    theta = np.asarray(theta, dtype=float)
    g = np.zeros_like(theta)
    for k in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        g[k] = (
            joint_logp(tp, i_raw, waner, cohort, splits, ignore_pcrpos, dense=dense)
            - joint_logp(tm, i_raw, waner, cohort, splits, ignore_pcrpos, dense=dense)
        ) / (2 * h)
    return g


def initial_theta() -> np.ndarray:
    """Transformed prior means (theta_init)."""

    def logit(x):
        return math.log(x / (1 - x))

    return np.array(
        [
            logit(0.02),  # p ~ 1/G-ish; any interior point will do
            math.log(2.0),
            math.log(1.0),
            logit(10.0 / 11.0),
            -2.0,
            math.log(2.0),
            logit(10.0 / 11.0),
            0.0,
            math.log(1.0),
            math.log(1.0),
            -2.0,
            -1.0,
            2.0,
            math.log(1.0),
            -1.0,
            2.0,
            math.log(1.0),
        ]
    )
