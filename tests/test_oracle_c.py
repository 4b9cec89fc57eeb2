"""The plain-C restatement (oracle/abd_oracle.c) against the NumPy oracle (pinned on the reference's vectors)."""
import numpy as np
import pytest

from abdpymc_amd import synthetic
from oracle import abd_oracle as O
from oracle import c_oracle
from tests.helpers import oracle_cohort_from_synth, random_sparse_cohort


def _state(coh, seed, rate=None):
    rng = np.random.default_rng(seed)
    rate = rate if rate is not None else 2.0 / coh.n_gaps
    i_raw = (rng.random((coh.n_gaps, coh.n_inds)) < rate).astype(np.int8)
    w = (rng.random(coh.n_inds) < 0.5).astype(np.int8)
    theta = synthetic.theta_init(coh.n_gaps) + 0.3 * rng.standard_normal(17)
    return theta, i_raw, w


@pytest.mark.parametrize("splits", [None, (9,), (7, 15), (0,), (20,)])
@pytest.mark.parametrize("ignore", [False, True])
@pytest.mark.parametrize("rate", [None, 0.6])
def test_c_oracle_dense(splits, ignore, rate):
    coh = oracle_cohort_from_synth(synthetic.make_cohort(31, 20, seed=5))
    theta, i_raw, w = _state(coh, 1, rate)
    co = c_oracle.COracle(coh, splits, ignore)
    lp, g, i = co.logp_dlogp(theta, i_raw, w, want_i=True)
    lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh, splits, ignore)
    i_ref, _, _ = O.deterministics(theta, i_raw, w, coh, splits, ignore)
    np.testing.assert_array_equal(i, i_ref)
    assert abs(lp - lp_ref) <= 1e-12 * abs(lp_ref)
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9 * np.abs(g_ref).max())


def test_c_oracle_sparse_and_threads():
    coh = random_sparse_cohort(40, 26, 800, 650, seed=3)
    theta, i_raw, w = _state(coh, 2)
    co = c_oracle.COracle(coh, (10,))
    a = co.logp_dlogp(theta, i_raw, w, nthreads=1)
    b = co.logp_dlogp(theta, i_raw, w, nthreads=4)
    ref = O.logp_dlogp(theta, i_raw, w, coh, (10,))
    assert abs(a[0] - ref[0]) <= 1e-12 * abs(ref[0])
    assert abs(b[0] - ref[0]) <= 1e-12 * abs(ref[0])
    np.testing.assert_allclose(a[1], ref[1], rtol=1e-9, atol=1e-9 * np.abs(ref[1]).max())
    np.testing.assert_allclose(b[1], ref[1], rtol=1e-9, atol=1e-9 * np.abs(ref[1]).max())
