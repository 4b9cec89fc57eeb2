"""
Pin the NumPy oracle against the reference's own known-answer tests (abdpymc/test_abd.py),
restated as data in tests/golden/reference_known_answers.json.
"""
import json
import os

import numpy as np
import pytest

from oracle import abd_oracle as O


@pytest.fixture(scope="module")
def K(golden_dir):
    with open(os.path.join(golden_dir, "reference_known_answers.json")) as f:
        return json.load(f)


def test_mask_future_infection_truth_table(K):
    for (i0, im3, im2, im1), want in K["mask_future_infection"]["cases"]:
        assert int(O.mask_future_infection(i0, im3, im2, im1)) == want


def test_mask_three_gaps(K):
    c = K["mask_three_gaps"]
    out = O.mask_three_gaps(np.array(c["input"]))
    assert out.dtype == np.int8
    assert out.shape == (5, 3)
    np.testing.assert_array_equal(out, c["expect"])


def test_mask_multiple_infections_exact(K):
    c = K["mask_multiple_infections"]
    np.testing.assert_array_equal(O.mask_multiple_infections(np.array(c["input"])), c["expect"])


def test_mask_multiple_infections_random_properties(K):
    c = K["mask_multiple_infections_random"]
    np.random.seed(c["seed"])
    arr = np.random.randint(0, 2, size=tuple(c["shape"]))
    out = O.mask_multiple_infections(arr)
    assert out.shape == (29, 50)
    assert out.sum(axis=0).max() == 1
    assert set(out.ravel()) == {0, 1}
    out2 = O.mask_multiple_infections_2_chunks(arr, split=c["split"])
    assert out2.shape == (29, 50)
    assert out2.sum(axis=0).max() == 2
    assert set(out2.ravel()) == {0, 1}


def test_mask_multiple_infections_chunks(K):
    c = K["mask_multiple_infections_2_chunks"]
    np.testing.assert_array_equal(
        O.mask_multiple_infections_2_chunks(np.array(c["input"]), c["split"]), c["expect"]
    )
    c = K["mask_multiple_infections_3_chunks"]
    np.testing.assert_array_equal(
        O.mask_multiple_infections_3_chunks(np.array(c["input"]), c["split0"], c["split1"]),
        c["expect"],
    )


def test_incorporate_pcrpos(K):
    c = K["incorporate_pcrpos"]
    np.testing.assert_array_equal(
        O.incorporate_pcrpos(np.array(c["i_raw"]), np.array(c["pcrpos"])), c["expect"]
    )


@pytest.mark.parametrize("name", ["two_chunks_incorporate_pcrpos_a", "two_chunks_incorporate_pcrpos_b"])
def test_two_chunks_incorporate(K, name):
    c = K[name]
    i_raw, pcr, s = np.array(c["i_raw"]), np.array(c["pcrpos"]), c["split"]
    out = np.concatenate(
        (O.incorporate_pcrpos(i_raw[:s], pcr[:s]), O.incorporate_pcrpos(i_raw[s:], pcr[s:]))
    )
    np.testing.assert_array_equal(out, c["expect"])


def test_two_chunks_constrain_infections(K):
    c = K["two_chunks_constrain_infections"]
    out = O.constrain_infections(np.array(c["i_raw"]), np.array(c["pcrpos"]), (c["split"],))
    np.testing.assert_array_equal(out, c["expect"])


def test_invlogistic(K):
    c = K["invlogistic"]
    kw = dict(a=c["a"], b=c["b"], d=c["d"])
    assert O.logistic(O.invlogistic(c["x"], **kw), **kw) == c["x"]


@pytest.mark.parametrize("name", ["temp_response_scalar_5x3", "temp_response_scalar_15x11"])
def test_temp_response_scalar(K, name):
    c = K[name]
    e = np.array(c["exposure"])
    np.testing.assert_almost_equal(O.temp_response_scalar_rho(e, c["temp"], c["rho"]), c["expect"])
    np.testing.assert_almost_equal(O.temp_response_scan(e, c["temp"], c["rho"]), c["expect"])


def test_temp_response_no_exposure(K):
    c = K["temp_response_no_exposure"]
    out = O.temp_response_scalar_rho(np.zeros(c["shape"]), c["temp"], c["rho"])
    np.testing.assert_almost_equal(out, np.zeros(c["shape"]))


def test_dense_equals_scan_seed42(K):
    c = K["temp_response_dense_vs_scan"]
    np.random.seed(c["seed"])
    for _ in range(c["repeats"]):
        n_gaps = np.random.randint(1, 20)
        n_inds = np.random.randint(1, 10)
        e = np.random.randint(0, 2, size=(n_gaps, n_inds))
        temp = np.random.uniform(0.1, 2.0)
        rho = np.random.uniform(0.1, 0.9)
        np.testing.assert_almost_equal(
            O.temp_response_scalar_rho(e, temp, rho), O.temp_response_scan(e, temp, rho)
        )


def test_temp_response_vector_rho(K):
    c = K["temp_response_vector_rho"]
    e = np.array(c["exposure"])
    rho = np.array(c["rho"])
    np.testing.assert_almost_equal(O.temp_response_vector_rho(e, c["temp"], rho), c["expect"])
    # quirk Q1: temp is ignored
    np.testing.assert_array_equal(
        O.temp_response_vector_rho(e, 123.0, rho), O.temp_response_vector_rho(e, 1.0, rho)
    )
    np.testing.assert_almost_equal(O.temp_response_scan(e, 1.0, rho), c["expect"])


def test_dense_equals_scan_large():
    rng = np.random.default_rng(0)
    e = (rng.random((200, 7)) < 0.05).astype(float)
    rho = rng.uniform(0.8, 0.99, size=7)
    a = O.temp_response_vector_rho(e, 1.0, rho)
    b = O.temp_response_scan(e, 1.0, rho)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)


def test_check_splits_messages():
    with pytest.raises(ValueError, match="split indexes must be positive"):
        O.check_splits((-1,))
    with pytest.raises(ValueError, match="ascending"):
        O.check_splits((5, 2))
    with pytest.raises(ValueError, match="largest split"):
        O.check_splits((3, 30), n_gaps=26)
    with pytest.raises(ValueError, match="not unique"):
        O.check_splits((3, 3))
    with pytest.raises(ValueError, match="must be ints"):
        O.check_splits((3.0,))
    O.check_splits((26,), n_gaps=26)  # Q6: == n_gaps is allowed
    O.check_splits(None)
