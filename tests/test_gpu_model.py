"""
The host mirror on the GPU: abdpymc_amd.model(...) -> compile_logp / logp_dlogp_function / deterministics
against the committed golden vectors and the live oracle, on the reference's own test cohort and default
cohort (BASELINE config 1: sparse observation lists, several dilutions per sample), plus the sampler / CLI.
"""
import json
import os

import numpy as np
import pytest

from abdpymc_amd.data import TiterData
from oracle import abd_oracle as O
from tests.test_data_loader import default_cohort

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _oracle_cohort(td):
    return O.Cohort(td.n_gaps, td.n_inds, np.asarray(td.vacs, dtype=np.int8), np.asarray(td.pcrpos, dtype=np.int8),
                    O.AntigenObs(*td.s.obs), O.AntigenObs(*td.n.obs))


def _close(lp, g, lp_ref, g_ref):
    assert abs(lp - lp_ref) <= RTOL * abs(lp_ref), (lp, lp_ref)
    scale = np.maximum(np.abs(g_ref), 1e-6 * np.abs(g_ref).max())
    assert (np.abs(np.asarray(g) - g_ref) / scale).max() <= RTOL


@pytest.fixture(scope="module")
def test_td(golden_dir):
    return TiterData.from_disk(os.path.join(golden_dir, "test_cohort"))


def test_golden_vectors_test_cohort(golden_dir, test_td):
    from abdpymc_amd.model import THETA_NAMES, model

    cases = json.load(open(os.path.join(golden_dir, "logp_golden.json")))["cases"]
    built = {}
    for c in cases:
        key = (tuple(c["splits"]), c["ignore_pcrpos"])
        if key not in built:
            built[key] = model(test_td, splits=tuple(c["splits"]) or None, ignore_pcrpos=c["ignore_pcrpos"])
        m = built[key]
        point = {n: np.asarray(v) for n, v in zip(THETA_NAMES, c["theta"])}
        point["i_raw"] = np.array(c["i_raw"], dtype=np.int64)
        point["ab_s_waner"] = np.array(c["waner"], dtype=np.int64)
        fn = m.logp_dlogp_function()
        fn.set_extra_values(point)
        lp, g = fn(m.ravel(point))
        _close(lp, g, c["logp"], np.array(c["grad"]))
        assert abs(m.compile_logp()(point) - c["logp"]) <= RTOL * abs(c["logp"])
        det = m.deterministics(point)
        np.testing.assert_array_equal(det["i"], np.array(c["i"], dtype=np.int8))  # integer pre-pass: bit-exact
        assert det["i"].dtype == np.int8 and det["i"].shape == (26, 10)
        np.testing.assert_allclose(det["ab_n_mu"][-1], c["mu_n_last"], rtol=1e-12)
        np.testing.assert_allclose(det["ab_s_mu"][-1], c["mu_s_last"], rtol=1e-12)
        assert abs(det["ab_n_mu"].sum() - c["mu_n_sum"]) <= 1e-10 * abs(c["mu_n_sum"])
        assert abs(det["ab_s_mu"].sum() - c["mu_s_sum"]) <= 1e-10 * abs(c["mu_s_sum"])


@pytest.mark.parametrize("splits", [None, (14,), (14, 20)])
def test_default_cohort_config1(golden_dir, splits):
    """N=1520, G=31, K_s=19508, K_n=16201: the sparse CSR kernel against the oracle and the plain-C port."""
    from abdpymc_amd.model import model
    from oracle import c_oracle

    td = default_cohort(golden_dir)
    coh = _oracle_cohort(td)
    m = model(td, splits=splits, n_chains=2)
    assert not m.ctx.is_dense
    rng = np.random.default_rng(7)
    pt = m.initial_point()
    q0 = m.ravel(pt)
    co = c_oracle.COracle(coh, splits)
    for chain in range(2):
        i_raw = (rng.random((td.n_gaps, td.n_inds)) < 1.0 / td.n_gaps).astype(np.int8)
        w = (rng.random(td.n_inds) < 0.5).astype(np.int8)
        theta = q0 + 0.3 * rng.standard_normal(17)
        m.ctx.set_discrete(chain, i_raw, w)
        lp, g = m.ctx.logp_dlogp(chain, theta)
        _close(lp, g, *O.logp_dlogp(theta, i_raw, w, coh, splits))
        _close(lp, g, *co.logp_dlogp(theta, i_raw, w))
    # the PyMC initial point (i_raw = 0, waner = 1) evaluates and is finite
    fn = m.logp_dlogp_function()
    fn.set_extra_values(pt)
    lp0, g0 = fn(q0)
    assert np.isfinite(lp0) and np.all(np.isfinite(g0))


def test_loglik_only_is_joint_minus_priors(test_td):
    from abdpymc_amd.model import model

    m = model(test_td)
    coh = _oracle_cohort(test_td)
    rng = np.random.default_rng(3)
    i_raw = (rng.random((26, 10)) < 0.1).astype(np.int8)
    w = (rng.random(10) < 0.5).astype(np.int8)
    theta = m.ravel(m.initial_point()) + 0.2 * rng.standard_normal(17)
    m.ctx.set_discrete(0, i_raw, w)
    ll, gl = m.ctx.loglik_dlogp(0, theta)
    lp, g = m.ctx.logp_dlogp(0, theta)
    lp0, g0 = O.prior_logp_grad(theta, 26, 260, int(i_raw.sum()), 10, int(w.sum()))
    assert abs((lp - ll) - lp0) <= 1e-9 * abs(lp)
    np.testing.assert_allclose(g - gl, g0, rtol=1e-9, atol=1e-9 * np.abs(g).max())
    assert gl[0] == 0 and gl[7] == 0 and gl[8] == 0 and gl[9] == 0


def test_gibbs_flip_path_matches_fresh_upload(test_td):
    """compile_logp()'s one-bit-changed fast path (device-side flip) == a fresh set_discrete."""
    from abdpymc_amd.model import model

    m = model(test_td, splits=(14, 20), n_chains=2)
    coh = _oracle_cohort(test_td)
    logp_fn = m.compile_logp(chain=0)
    pt = m.initial_point()
    rng = np.random.default_rng(5)
    for _ in range(25):
        k = int(rng.integers(0, 270))
        if k < 260:
            pt["i_raw"].ravel()[k] ^= 1
        else:
            pt["ab_s_waner"][k - 260] ^= 1
        got = logp_fn(pt)
        ref = O.logp_dlogp(m.ravel(pt), pt["i_raw"], pt["ab_s_waner"], coh, (14, 20))[0]
        assert abs(got - ref) <= RTOL * abs(ref)


def test_sampler_smoke_and_cli(tmp_path, golden_dir, test_td):
    from abdpymc_amd import cli
    from abdpymc_amd.model import model
    from abdpymc_amd.sampler import sample

    m = model(test_td, n_chains=2)
    res = sample(m, tune=15, draws=10, chains=2, seed=1)
    res_host = sample(m, tune=4, draws=3, chains=1, seed=2, device_gibbs=False)  # host-driven flips still work
    assert res_host["i"].shape == (1, 3, 26, 10) and np.all(np.isfinite(res_host["stat_lp"]))
    assert res["p"].shape == (2, 10) and res["i"].shape == (2, 10, 26, 10) and res["ab_s_mu"].shape == (2, 10, 26, 10)
    assert np.all(np.isfinite(res["stat_lp"])) and np.all((res["p"] > 0) & (res["p"] < 1))
    # recorded Deterministic i is the constrained i_raw of the same draw
    coh = _oracle_cohort(test_td)
    i_ref = O.constrain_infections(res["i_raw"][1, -1], np.asarray(test_td.pcrpos).T)
    np.testing.assert_array_equal(res["i"][1, -1], i_ref)
    out = tmp_path / "post"
    rc = cli.main(["--tune", "5", "--draws", "4", "--cores", "1", "--ititers_data", os.path.join(golden_dir, "test_cohort"),
                   "--split_delta", "--netcdf", str(out)])
    assert rc == 0
    files = list(tmp_path.iterdir())
    assert len(files) == 1
    if files[0].suffix == ".npz":
        z = np.load(files[0])
        # --cores 1 -> two chains, as pm.sample(cores=1) runs max(2, cores)
        assert z["ab_n_mu"].shape == (2, 4, 26, 10) and z["it_s_sigma"].shape == (2, 4)


def test_both_callables_on_one_chain_slot_stay_coherent(test_td):
    """INTEGRATION.md wires compile_logp() and logp_dlogp_function() to the same chain slot, as PyMC's compound step
    does: NUTS re-uploads the accepted state through set_extra_values while BinaryGibbsMetropolis flips single bits
    through the point function.  A sweep that ends on a REJECTED proposal leaves the point function's mirror on the
    flipped state; the next sweep must not trust it once NUTS has rewritten the slot."""
    import abdpymc_amd

    coh = _oracle_cohort(test_td)
    m = abdpymc_amd.model(test_td, splits=(14,))
    logp_fn, fn = m.compile_logp(), m.logp_dlogp_function()
    rng = np.random.default_rng(11)
    pt = m.initial_point()
    pt["i_raw"] = (rng.random((m.n_gaps, m.n_inds)) < 0.06).astype(np.int64)
    theta = m.ravel(pt)

    def ref(i_raw, w):
        return O.logp_dlogp(theta, i_raw, w, coh, (14,))[0]

    fn.set_extra_values(pt)
    assert abs(fn(theta)[0] - ref(pt["i_raw"], pt["ab_s_waner"])) <= RTOL * abs(ref(pt["i_raw"], pt["ab_s_waner"]))
    accepted = {k: np.array(v, copy=True) for k, v in pt.items()}
    for sweep in range(4):
        # a Gibbs "sweep" of a few proposals, the LAST one rejected (the caller keeps `accepted`, the device and the
        # point function's mirror are left on the rejected proposal)
        for k in range(3):
            prop = {kk: np.array(v, copy=True) for kk, v in accepted.items()}
            f = int(rng.integers(m.n_gaps * m.n_inds))
            prop["i_raw"].ravel()[f] ^= 1
            want = ref(prop["i_raw"], prop["ab_s_waner"])
            assert abs(float(logp_fn(prop)) - want) <= RTOL * abs(want)
            if k < 2:
                accepted = prop
        # NUTS: the accepted state goes up through the OTHER callable, then leapfrogs
        fn.set_extra_values(accepted)
        want = ref(accepted["i_raw"], accepted["ab_s_waner"])
        assert abs(fn(theta)[0] - want) <= RTOL * abs(want)
        # next sweep's first proposal differs from the stale mirror by exactly one bit more than it thinks
        prop = {kk: np.array(v, copy=True) for kk, v in accepted.items()}
        prop["ab_s_waner"][sweep] ^= 1
        want = ref(prop["i_raw"], prop["ab_s_waner"])
        assert abs(float(logp_fn(prop)) - want) <= RTOL * abs(want)
        fn.set_extra_values(accepted)
    # a device sweep rewrites the slot behind both callables' backs
    m.ctx.gibbs_sweep([0], theta[None], seed=3, sweep=0)
    i_dev, w_dev = m.ctx.get_discrete(0)
    assert abs(float(logp_fn(accepted)) - ref(accepted["i_raw"], accepted["ab_s_waner"])) <= RTOL * abs(ref(accepted["i_raw"], accepted["ab_s_waner"]))
    assert logp_fn.mirror.uploads >= 2 and logp_fn.mirror.flips >= 8
    m.close()


def test_pytensor_op_core_constrained_gradient_and_uploads(test_td):
    """What AbdDataLogp.perform computes (PyTensor itself is not installed here): the data term as a function of the 13
    CONSTRAINED parameters -- its gradient against central finite differences in that space, the value against the
    oracle's joint logp minus priors -- and the discrete state is uploaded once, not per call."""
    from abdpymc_amd.pytensor_op import _Lazy

    lazy = _Lazy(test_td, (14, 20), False)
    m = lazy.model()
    rng = np.random.default_rng(5)
    i_raw = (rng.random((m.n_gaps, m.n_inds)) < 0.05).astype(np.int8)
    waner = (rng.random(m.n_inds) < 0.5).astype(np.int8)
    params = np.array([2.1, 0.9, 0.88, -1.9, 1.8, 0.93, -2.2, -1.2, 1.7, 0.35, -0.8, 2.2, 0.4])
    ll, g = lazy.loglik_dlogp_constrained(params, i_raw, waner)
    for k in range(13):
        h = 1e-6 * max(1.0, abs(params[k]))
        up, dn = params.copy(), params.copy()
        up[k] += h
        dn[k] -= h
        fd = (lazy.loglik_dlogp_constrained(up, i_raw, waner)[0] - lazy.loglik_dlogp_constrained(dn, i_raw, waner)[0]) / (2 * h)
        assert abs(fd - g[k]) <= 2e-6 * max(abs(g[k]), 1e-3 * np.abs(g).max()), (k, fd, g[k])
    # NUTS leapfrogs: same discrete arrays every call -> one upload in all
    assert lazy.mirror().uploads == 1 and lazy.mirror().flips == 0 and lazy.mirror().hits == 26
    # a Gibbs proposal: one bit differs -> one flip, no upload
    i2 = i_raw.copy()
    i2[3, 2] ^= 1
    ll2, _ = lazy.loglik_dlogp_constrained(params, i2, waner)
    assert lazy.mirror().uploads == 1 and lazy.mirror().flips == 1
    from abdpymc_amd.pytensor_op import _theta_from_constrained

    coh = _oracle_cohort(test_td)
    theta = _theta_from_constrained(params, m.n_gaps)
    for ir, val in ((i_raw, ll), (i2, ll2)):
        joint = O.logp_dlogp(theta, ir, waner, coh, (14, 20))[0]
        prior = O.prior_logp_grad(theta, m.n_gaps, m.n_gaps * m.n_inds, int(ir.sum()), m.n_inds, int(waner.sum()))[0]
        assert abs(val - (joint - prior)) <= 1e-9 * abs(joint)
    m.close()


def test_pymc_shaped_step_over_the_device_sweep(test_td):
    """GibbsSweepStep.astep(point) -> (point, stats), the core of the step that stands in for pm.BinaryGibbsMetropolis
    (INTEGRATION.md level 2): each call is exactly one abd_gibbs_sweep of the slot at the point's continuous values --
    same bits as driving Context.gibbs_sweep + get_discrete by hand on a twin context -- and it uploads the discrete
    state only when the device does not already hold it; NUTS's callable on the same slot then sees the swept state."""
    import abdpymc_amd
    from abdpymc_amd.pymc_step import GibbsSweepStep

    coh = _oracle_cohort(test_td)
    m = abdpymc_amd.model(test_td, splits=(14,))
    twin = abdpymc_amd.model(test_td, splits=(14,))
    rng = np.random.default_rng(5)
    pt = m.initial_point()
    pt["i_raw"] = (rng.random((m.n_gaps, m.n_inds)) < 0.05).astype(np.int64)
    theta = m.ravel(pt)
    step = GibbsSweepStep(m, chain=0, seed=77)
    fn = m.logp_dlogp_function()
    twin.ctx.set_discrete(0, pt["i_raw"], pt["ab_s_waner"])
    for k in range(4):
        new_pt, stats = step.astep(pt)
        acc, prop = twin.ctx.gibbs_sweep([0], theta[None], seed=77, sweep=k)
        i_ref, w_ref = twin.ctx.get_discrete(0)
        np.testing.assert_array_equal(new_pt["i_raw"], i_ref)
        np.testing.assert_array_equal(new_pt["ab_s_waner"], w_ref)
        assert stats == {"accepted": int(acc[0]), "proposed": int(prop[0])} and stats["proposed"] > 0
        assert new_pt["i_raw"].dtype == pt["i_raw"].dtype and set(new_pt) == set(pt)
        # the continuous part of the point is passed through untouched
        assert all(np.array_equal(new_pt[n], pt[n]) for n in m.continuous_value_vars)
        # NUTS on the same slot: its callable uploads the swept state (the sweep bumped the slot's generation) and
        # evaluates there
        fn.set_extra_values(new_pt)
        want = O.logp_dlogp(theta, new_pt["i_raw"], new_pt["ab_s_waner"], coh, (14,))[0]
        assert abs(fn(theta)[0] - want) <= RTOL * abs(want)
        pt = new_pt
    # handing back exactly what the step returned costs no upload: one at the start, one after each of NUTS's rewrites
    assert step.mirror.uploads == 4 and step.n_sweeps == 4
    # ... and none at all when nobody else touches the slot in between
    solo = GibbsSweepStep(twin, chain=0, seed=1)
    p2 = dict(pt)
    for k in range(3):
        p2, _ = solo.astep(p2)
    assert solo.mirror.uploads == 1 and solo.mirror.hits == 2
    # copies of one step method (what pm.sample hands its chains) must not share a random stream: without a seed each copy
    # draws its own at its first sweep, or takes the generator PyMC gives the chain's step
    import copy

    proto = GibbsSweepStep(twin, chain=0)
    a_step, b_step = copy.copy(proto), copy.copy(proto)
    assert proto.seed is None
    a_step.astep(dict(pt))
    b_step.astep(dict(pt))
    assert a_step.seed is not None and a_step.seed != b_step.seed and proto.seed is None
    c_step, d_step = copy.copy(proto), copy.copy(proto)
    c_step.set_rng(np.random.default_rng(1))
    d_step.set_rng(np.random.default_rng(1))
    assert c_step.seed == d_step.seed  # (the same generator state gives the same stream: reproducible runs)
    m.close()
    twin.close()
