"""
Joint logp / dlogp of the oracle: PARITY UNPINNED against the reference (no reference test evaluates
model.logp; PyMC is not installed).  What is checked instead:
  * literal dense (G,G,N) form == recurrence form,
  * closed-form priors == scipy.stats densities + explicit Jacobians,
  * analytic gradient == central finite differences (rel 1e-6).
"""
import math

import numpy as np
import pytest
from scipy import stats

from abdpymc_amd import synthetic
from oracle import abd_oracle as O
from tests.helpers import oracle_cohort_from_synth, random_sparse_cohort

SPLITS = [None, (), (9,), (7, 15), (0,), (20,), (0, 20)]


@pytest.fixture(scope="module")
def small():
    sc = synthetic.make_cohort(n_inds=23, n_gaps=20, seed=3)
    return oracle_cohort_from_synth(sc)


def _state(coh, seed):
    rng = np.random.default_rng(seed)
    i_raw = (rng.random((coh.n_gaps, coh.n_inds)) < 0.15).astype(np.int8)
    w = (rng.random(coh.n_inds) < 0.5).astype(np.int8)
    theta = synthetic.theta_init(coh.n_gaps) + 0.3 * rng.standard_normal(17)
    return theta, i_raw, w


@pytest.mark.parametrize("splits", SPLITS)
@pytest.mark.parametrize("ignore", [False, True])
def test_dense_equals_recurrence(small, splits, ignore):
    theta, i_raw, w = _state(small, 1)
    a = O.joint_logp(theta, i_raw, w, small, splits, ignore, dense=True)
    b = O.joint_logp(theta, i_raw, w, small, splits, ignore, dense=False)
    c, _ = O.logp_dlogp(theta, i_raw, w, small, splits, ignore)
    assert abs(a - b) <= 1e-12 * abs(a)
    assert abs(a - c) <= 1e-12 * abs(a)


def test_priors_match_scipy():
    rng = np.random.default_rng(5)
    G, N = 31, 40
    for _ in range(5):
        t = synthetic.theta_init(G) + 0.5 * rng.standard_normal(17)
        n1, m1 = 57, 22
        c = O.constrained(t)
        want = 0.0
        # p: Beta(1, G-1) + logodds Jacobian; Bernoulli(i_raw | p)
        want += stats.beta.logpdf(c["p"], 1, G - 1) + math.log(c["p"]) + math.log1p(-c["p"])
        want += n1 * math.log(c["p"]) + (G * N - n1) * math.log1p(-c["p"])
        for k, mu, sd in ((1, 2.0, 0.5), (2, 1.0, 0.5), (5, 2.0, 0.5), (8, 1.0, 0.5), (9, 1.0, 0.5)):
            al, be = mu**2 / sd**2, mu / sd**2
            want += stats.gamma.logpdf(math.exp(t[k]), al, scale=1 / be) + t[k]
        for k in (3, 6):
            r = O._sigmoid(t[k])
            want += stats.beta.logpdf(r, 10, 1) + math.log(r) + math.log1p(-r)
        q = c["p_waner"]
        want += stats.beta.logpdf(q, 1, 1) + math.log(q) + math.log1p(-q)
        want += m1 * math.log(q) + (N - m1) * math.log1p(-q)
        want += stats.norm.logpdf(t[4], -2, 1) + stats.norm.logpdf(t[10], -2, 1)
        for o in (11, 14):
            want += stats.norm.logpdf(t[o], -1, 0.5) + stats.norm.logpdf(t[o + 1], 2, 0.5)
            want += stats.expon.logpdf(math.exp(t[o + 2])) + t[o + 2]
        lit = O.prior_logp_literal(t, G, G * N, n1, N, m1)
        closed, _ = O.prior_logp_grad(t, G, G * N, n1, N, m1)
        assert abs(lit - want) <= 1e-12 * abs(want)
        assert abs(closed - want) <= 1e-12 * abs(want)


def test_prior_gradient_fd():
    rng = np.random.default_rng(6)
    G, N = 26, 10
    t = synthetic.theta_init(G) + 0.4 * rng.standard_normal(17)
    _, g = O.prior_logp_grad(t, G, G * N, 9, N, 4)
    h = 1e-6
    for k in range(17):
        tp, tm = t.copy(), t.copy()
        tp[k] += h
        tm[k] -= h
        fd = (O.prior_logp_grad(tp, G, G * N, 9, N, 4)[0] - O.prior_logp_grad(tm, G, G * N, 9, N, 4)[0]) / (2 * h)
        assert abs(fd - g[k]) <= 1e-6 * max(1.0, abs(g[k]))


@pytest.mark.parametrize("splits", [None, (9,), (7, 15)])
def test_gradient_matches_finite_differences(small, splits):
    theta, i_raw, w = _state(small, 2)
    lp, g = O.logp_dlogp(theta, i_raw, w, small, splits)
    fd = O.finite_difference_grad(theta, i_raw, w, small, splits, h=1e-5, dense=False)
    assert g[8] != 0 and g[9] != 0  # tempinf / tempvac: prior-only (Q1) but not zero
    np.testing.assert_allclose(g, fd, rtol=1e-6, atol=1e-6 * np.abs(g).max())


def test_gradient_sparse_cohort_fd():
    coh = random_sparse_cohort(17, 26, 300, 250, seed=4)
    theta, i_raw, w = _state(coh, 3)
    lp, g = O.logp_dlogp(theta, i_raw, w, coh, (10,))
    fd = O.finite_difference_grad(theta, i_raw, w, coh, (10,), h=1e-5, dense=True)
    np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-6 * np.abs(g).max())


def test_q1_temp_ignored_in_s(small):
    """tempinf / tempvac only enter through their Gamma priors (abd.py:272-274)."""
    theta, i_raw, w = _state(small, 7)
    _, _, mu_s0 = O.deterministics(theta, i_raw, w, small, dense=True)
    t2 = theta.copy()
    t2[8] += 1.0
    t2[9] -= 0.7
    _, _, mu_s1 = O.deterministics(t2, i_raw, w, small, dense=True)
    np.testing.assert_array_equal(mu_s0, mu_s1)


def test_q2_bernoulli_on_raw(small):
    """Bits the masks erase still pay prior cost: logp changes although i does not."""
    theta, i_raw, w = _state(small, 8)
    i_raw = i_raw.copy()
    i_raw[:] = 0
    i_raw[3, 0] = 1
    a = O.joint_logp(theta, i_raw, w, small, dense=False)
    i2 = i_raw.copy()
    i2[4, 0] = 1  # masked by the 3-gap rule
    ia, _, _ = O.deterministics(theta, i_raw, w, small, ignore_pcrpos=True)
    ib, _, _ = O.deterministics(theta, i2, w, small, ignore_pcrpos=True)
    np.testing.assert_array_equal(ia, ib)
    b = O.joint_logp(theta, i2, w, small, ignore_pcrpos=True, dense=False)
    a = O.joint_logp(theta, i_raw, w, small, ignore_pcrpos=True, dense=False)
    p = O.constrained(theta)["p"]
    assert abs((b - a) - (math.log(p) - math.log1p(-p))) < 1e-9
