"""
Synthetic cohorts for the benchmark configurations (BASELINE.json configs 2-5; recipe SURVEY 8d).

Dense panel: exactly one S and one N OD reading in every (gap, ind) cell, so K_s = K_n = G*N.  Event
rates mirror the reference's default cohort (1046 doses and 722 PCR+ over 1520 people x 31 months);
truth dynamics / ELISA parameters are the defaults of the reference's forward simulator
(abdpymc/simulation.py:76-78, 104-108) pushed through the *model's* equations (abd.py:309-393).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

SEED = 20231202


@dataclasses.dataclass
class SyntheticCohort:
    n_gaps: int
    n_inds: int
    vacs: np.ndarray  # (N, G) int8
    pcrpos: np.ndarray  # (N, G) int8
    # observation lists, sorted by (ind, gap):  k = j*G + g
    idx_gap: np.ndarray  # (G*N,) int32
    idx_ind: np.ndarray
    x_s: np.ndarray  # log_dilution
    y_s: np.ndarray  # od
    x_n: np.ndarray
    y_n: np.ndarray
    i_true: np.ndarray = None  # (G, N) the infections the ODs were simulated from (parameter-recovery tests)

    @property
    def s_obs(self):
        return self.idx_gap, self.idx_ind, self.x_s, self.y_s

    @property
    def n_obs(self):
        return self.idx_gap, self.idx_ind, self.x_n, self.y_n


def _mask_three_gaps(i0: np.ndarray) -> np.ndarray:
    """out[t] = in[t] unless out[t-1]|out[t-2]|out[t-3]  (abd.py:560-601), vectorised over individuals."""
    out = np.zeros_like(i0)
    G = i0.shape[0]
    for t in range(G):
        blocked = np.zeros(i0.shape[1], dtype=bool)
        for k in (1, 2, 3):
            if t - k >= 0:
                blocked |= out[t - k] != 0
        out[t] = np.where(blocked, 0, i0[t])
    return out


def theta_init(n_gaps: int) -> np.ndarray:
    """Transformed prior means."""

    def logit(p):
        return math.log(p / (1.0 - p))

    return np.array(
        [
            logit(1.0 / n_gaps),  # E[Beta(1, G-1)] = 1/G
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
            0.0,
            -1.0,
            2.0,
            0.0,
        ]
    )


# what make_cohort simulates from (simulation.py:76-78, 104-108 defaults pushed through the model's equations)
TRUTH = dict(ab_n_init=-2.0, ab_s_init=-2.0, ab_n_perm=2.0, ab_s_perm=2.0, ab_n_temp=1.5, ab_n_rho=0.95, ab_s_rho=0.95,
             it_n_b=-2.2, it_s_b=-2.2, it_n_d=1.6, it_s_d=1.6, it_n_sigma=0.1, it_s_sigma=0.1)


def truth_theta(n_gaps: int) -> np.ndarray:
    """The unconstrained 17-vector at the simulation's own parameters (a converged chain's neighbourhood)."""

    def logit(p):
        return math.log(p / (1.0 - p))

    T = TRUTH
    return np.array([logit(1.0 / n_gaps), math.log(T["ab_n_perm"]), math.log(T["ab_n_temp"]), logit(T["ab_n_rho"]), T["ab_n_init"],
                     math.log(T["ab_s_perm"]), logit(T["ab_s_rho"]), logit(0.99), 0.0, 0.0, T["ab_s_init"],
                     T["it_n_b"], T["it_n_d"], math.log(T["it_n_sigma"]), T["it_s_b"], T["it_s_d"], math.log(T["it_s_sigma"])])


def make_cohort(n_inds: int, n_gaps: int, seed: int = SEED) -> SyntheticCohort:
    rng = np.random.default_rng(seed)
    G, N = n_gaps, n_inds
    vacs = (rng.random((N, G)) < 0.7 / G).astype(np.int8)
    pcrpos = (rng.random((N, G)) < 0.5 / G).astype(np.int8)
    i0 = ((rng.random((G, N)) < 1.0 / G) | (pcrpos.T != 0)).astype(np.int8)
    i_true = _mask_three_gaps(i0).astype(float)
    v = vacs.T.astype(float)
    # truth dynamics through the model's equations (unit S boosts, quirk Q1)
    init, perm, temp_n, wane = -2.0, 2.0, 1.5, 0.95
    tn = np.zeros((G, N))
    ts = np.zeros((G, N))
    pn = np.zeros(N)
    ps = np.zeros(N)
    for g in range(G):
        pn = pn * wane + temp_n * i_true[g]
        ps = ps * wane + i_true[g] + v[g]
        tn[g], ts[g] = pn, ps
    mu_n = init + perm * (np.cumsum(i_true, axis=0) > 0) + tn
    mu_s = init + perm * (np.cumsum(i_true + v, axis=0) > 0) + ts
    b, d, sd = -2.2, 1.6, 0.1
    # observation order: sorted by (ind, gap)
    idx_ind = np.repeat(np.arange(N, dtype=np.int32), G)
    idx_gap = np.tile(np.arange(G, dtype=np.int32), N)

    def od(mu):
        x = rng.choice(np.array([0.0, 2.0, 4.0]), size=G * N)
        a = np.ascontiguousarray(mu.T).ravel()  # = mu[idx_gap, idx_ind]: observation k = j * G + g reads mu[g, j]
        y = d / (1.0 + np.exp(-b * (x - a))) + sd * rng.standard_normal(G * N)
        return x, y

    x_s, y_s = od(mu_s)
    x_n, y_n = od(mu_n)
    return SyntheticCohort(G, N, vacs, pcrpos, idx_gap, idx_ind, x_s, y_s, x_n, y_n, i_true.astype(np.int8))


def make_chain_state(n_inds: int, n_gaps: int, chain: int, seed: int = SEED):
    """Per-chain discrete state: i_raw (G, N) ~ Bernoulli(1/G), waner (N,) ~ Bernoulli(1/2)."""
    rng = np.random.default_rng([seed, 1000 + chain])
    i_raw = (rng.random((n_gaps, n_inds)) < 1.0 / n_gaps).astype(np.int8)
    waner = (rng.random(n_inds) < 0.5).astype(np.int8)
    return i_raw, waner


def make_thetas(n_gaps: int, n: int, chain: int, seed: int = SEED, scale: float = 0.3) -> np.ndarray:
    """Evaluation points theta_init + 0.3 N(0, I): a fresh theta for every call."""
    rng = np.random.default_rng([seed, 2000 + chain])
    return theta_init(n_gaps)[None, :] + scale * rng.standard_normal((n, 17))
