"""
The reference's own known-answer vectors (abdpymc/test_abd.py, restated as data in
tests/golden/reference_known_answers.json) fed STRAIGHT through the HIP path: a context is built from each case,
`abd_deterministics` returns the Deterministics `i`, `ab_n_mu`, `ab_s_mu` (abd.py:649/667, 341, 389-391) and they
are compared with the arrays the reference's tests hold -- no oracle in between, except where a case tests one
function of a pipeline the library only runs whole (stated at each test).
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import abd_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K(golden_dir):
    with open(os.path.join(golden_dir, "reference_known_answers.json")) as f:
        return json.load(f)


def _logit(p):
    return math.log(p / (1.0 - p))


def _theta(rho_n=0.5, temp_n=1.0, rho_s=0.5, perm_n=2.0, perm_s=3.0, init_n=-2.0, init_s=-1.5):
    t = O.initial_theta().copy()
    t[1], t[2], t[3], t[4] = math.log(perm_n), math.log(temp_n), _logit(rho_n), init_n
    t[5], t[6], t[10] = math.log(perm_s), _logit(rho_s), init_s
    return t


def _det(i_raw, vacs=None, pcrpos=None, splits=None, waner=None, theta=None):
    """(i, ab_n_mu, ab_s_mu) of the HIP path for a (G, N) i_raw, (G, N) vacs / pcrpos as the reference's tests write them"""
    from abdpymc_amd._native import Context

    i_raw = np.asarray(i_raw)
    G, N = i_raw.shape
    vacs = np.zeros((G, N), dtype=np.int8) if vacs is None else np.asarray(vacs)
    pcr = np.zeros((G, N), dtype=np.int8) if pcrpos is None else np.asarray(pcrpos)
    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), np.zeros(0))
    ctx = Context(G, N, empty, empty, vacs.T.copy(), pcr.T.copy(), splits=splits)  # the library takes (N, G) like TiterData
    ctx.set_discrete(0, i_raw, np.ones(N, dtype=np.int8) if waner is None else waner)
    out = ctx.deterministics(0, _theta() if theta is None else theta)
    ctx.close()
    return out


def test_mask_three_gaps_through_hip(K):
    """test_abd.py:119-146: OneTimeChunk with no PCR+ is mask_three_gaps alone"""
    c = K["mask_three_gaps"]
    i, _, _ = _det(c["input"])
    np.testing.assert_array_equal(i, c["expect"])


def test_two_chunks_constrain_infections_through_hip(K):
    """test_abd.py:560-617: the whole constrain_infections of TwoTimeChunks"""
    c = K["two_chunks_constrain_infections"]
    i, _, _ = _det(c["i_raw"], pcrpos=c["pcrpos"], splits=(c["split"],))
    np.testing.assert_array_equal(i, c["expect"])


def test_mask_multiple_infections_through_hip(K):
    """test_abd.py:199-206, 278-366.  The library runs mask_multiple_infections only inside constrain_infections, i.e.
    followed by incorporate_pcrpos (no PCR+ here: identity) and mask_three_gaps: the reference's expected array is
    passed through mask_three_gaps (pinned on its own above); columns it leaves untouched are compared as they stand."""
    cases = [("mask_multiple_infections", None), ("mask_multiple_infections_2_chunks", "split"),
             ("mask_multiple_infections_3_chunks", ("split0", "split1"))]
    for name, sk in cases:
        c = K[name]
        x = np.array(c["input"])
        G = x.shape[0]
        # one chunk through the multi-chunk code: a split at n_gaps (allowed, abd.py:615) leaves an empty second chunk
        splits = (G,) if sk is None else ((c[sk],) if isinstance(sk, str) else tuple(c[s] for s in sk))
        i, _, _ = _det(x, splits=splits)
        want = np.array(c["expect"])
        np.testing.assert_array_equal(i, O.mask_three_gaps(want))
        direct = [j for j in range(x.shape[1]) if np.array_equal(O.mask_three_gaps(want[:, j:j + 1]), want[:, j:j + 1])]
        assert direct, name
        np.testing.assert_array_equal(i[:, direct], want[:, direct])


def test_incorporate_pcrpos_through_hip(K):
    """test_abd.py:370-413 (one chunk's incorporate_pcrpos), run as the multi-chunk pipeline with a split at n_gaps"""
    c = K["incorporate_pcrpos"]
    x = np.array(c["i_raw"])
    assert np.array_equal(O.mask_multiple_infections(x), x)  # the case has at most one infection per column
    i, _, _ = _det(x, pcrpos=c["pcrpos"], splits=(x.shape[0],))
    np.testing.assert_array_equal(i, O.mask_three_gaps(np.array(c["expect"])))


@pytest.mark.parametrize("name", ["temp_response_scalar_5x3", "temp_response_scalar_15x11"])
def test_temp_response_scalar_rho_through_hip(K, name):
    """test_abd.py:770-795, 797-974.  Two routes, both compared with the reference's expected array to its 7 decimals:
    (S) the exposures enter as vaccinations, which no mask touches (unit boosts, abd.py:272: the case has temp = 1);
    (N) infections go through mask_three_gaps, so the exposure matrix is cut into four layers of rows g = k mod 4 (each
    survives the mask unchanged) and the layers' responses are added -- the response is linear in the exposures."""
    c = K[name]
    e = np.array(c["exposure"])
    want = np.array(c["expect"])
    G, N = e.shape
    th = _theta(rho_n=c["rho"], temp_n=c["temp"], rho_s=c["rho"])
    zeros = np.zeros_like(e)
    _, _, mu_s = _det(zeros, vacs=e, theta=th)
    cum = (np.cumsum(e, axis=0) > 0)
    np.testing.assert_almost_equal(mu_s - th[10] - math.exp(th[5]) * cum, want, decimal=c["decimal"])
    total = np.zeros((G, N))
    for k in range(4):
        layer = np.zeros_like(e)
        layer[k::4] = e[k::4]
        i, mu_n, _ = _det(layer, theta=th)
        np.testing.assert_array_equal(i, layer)
        total += mu_n - th[4] - math.exp(th[1]) * (np.cumsum(layer, axis=0) > 0)
    np.testing.assert_almost_equal(total, want, decimal=c["decimal"])


def test_temp_response_no_exposure_through_hip(K):
    """test_abd.py:976-985: no exposure, no response"""
    c = K["temp_response_no_exposure"]
    G, N = c["shape"]
    th = _theta(rho_n=c["rho"], temp_n=c["temp"], rho_s=c["rho"])
    i, mu_n, mu_s = _det(np.zeros((G, N), dtype=np.int8), theta=th)
    assert not i.any()
    np.testing.assert_array_equal(mu_n, np.full((G, N), th[4]))
    np.testing.assert_array_equal(mu_s, np.full((G, N), th[10]))


def test_temp_response_vector_rho_through_hip(K):
    """test_abd.py:1013-1040: one rho per individual.  The model only ever has two values of rho_j (abd.py:374), so
    column k is evaluated with rho_s = rho[k]; a non-waner column must come out with rho_j = 1 (a plain cumulative sum)."""
    c = K["temp_response_vector_rho"]
    e = np.array(c["exposure"])
    want = np.array(c["expect"])
    G, N = e.shape
    for k, rho in enumerate(c["rho"]):
        th = _theta(rho_s=rho)
        waner = np.ones(N, dtype=np.int8)
        waner[(k + 1) % N] = 0
        _, _, mu_s = _det(np.zeros_like(e), vacs=e, theta=th, waner=waner)
        resp = mu_s - th[10] - math.exp(th[5]) * (np.cumsum(e, axis=0) > 0)
        np.testing.assert_almost_equal(resp[:, k], want[:, k], decimal=c["decimal"])
        np.testing.assert_almost_equal(resp[:, (k + 1) % N], np.cumsum(e[:, (k + 1) % N]), decimal=12)
