"""The NUTS state machine of the native sampler (abd_nuts.hpp, HIP-free) on a closed-form target, CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("nuts") / "libnuts_harness.so"
    # (-Bsymbolic: the harness must run ITS copy of abd_nuts.hpp's inline functions -- the product library, loaded RTLD_GLOBAL by
    # other tests of the same process, exports the same vague-linkage symbols compiled by another compiler with other contraction)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility-inlines-hidden", "-Wl,-Bsymbolic", "-I", os.path.join(ROOT, "abdpymc_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "nuts_harness.cpp"), "-o", str(out)])
    lib = C.CDLL(str(out))
    dp = C.POINTER(C.c_double)
    lib.nuts_harness_run.argtypes = [dp, dp, dp, C.c_longlong, C.c_longlong, C.c_ulonglong, C.c_int, C.c_int, dp, dp]
    lib.nuts_harness_run.restype = C.c_int

    def run(mean, sd, tune, draws, seed, chains, prec=None, dense=False):
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        sd = np.ascontiguousarray(sd, dtype=np.float64)
        pp = None
        if prec is not None:
            prec = np.ascontiguousarray(prec, dtype=np.float64)
            pp = prec.ctypes.data_as(dp)
        q = np.empty((chains, draws, 17))
        st = np.empty((chains, draws, 6))
        rc = lib.nuts_harness_run(mean.ctypes.data_as(dp), sd.ctypes.data_as(dp), pp, tune, draws, seed, chains, int(dense),
                                  q.ctypes.data_as(dp), st.ctypes.data_as(dp))
        assert rc == 0
        return q, st

    run.lib_path = str(out)
    return run


def _target():
    rng = np.random.default_rng(5)
    mean = rng.normal(size=17) * 3
    sd = np.exp(rng.uniform(np.log(0.01), np.log(10.0), size=17))  # three decades of scales: needs the metric
    return mean, sd


def test_moments_of_a_badly_scaled_normal(harness):
    mean, sd = _target()
    q, st = harness(mean, sd, 1000, 2000, 11, 4)
    assert not st[..., 5].any()  # no divergences on a normal
    flat = q.reshape(-1, 17)
    # effective sample size of NUTS on a normal is of the order of the draw count; allow 5 standard errors at ESS = n/4
    se = sd / np.sqrt(flat.shape[0] / 4)
    assert np.all(np.abs(flat.mean(0) - mean) < 5 * se), (flat.mean(0) - mean) / se
    assert np.all(np.abs(flat.std(0) / sd - 1) < 0.12), flat.std(0) / sd
    acc = st[..., 3].mean()
    assert 0.7 < acc < 0.95, acc
    # with an adapted metric the target is isotropic: trees stay short
    assert st[..., 1].mean() < 4.5, st[..., 1].mean()
    # logp stat is the target's logp at the draw
    lp = -0.5 * (((q - mean) / sd) ** 2).sum(-1)
    np.testing.assert_allclose(st[..., 0], lp, rtol=1e-10, atol=1e-10)


def test_reproducible_and_streams_differ(harness):
    mean, sd = _target()
    q1, _ = harness(mean, sd, 200, 100, 3, 2)
    q2, _ = harness(mean, sd, 200, 100, 3, 2)
    q3, _ = harness(mean, sd, 200, 100, 4, 2)
    np.testing.assert_array_equal(q1, q2)
    assert not np.array_equal(q1, q3)
    assert not np.array_equal(q1[0], q1[1])


def test_step_size_settles_near_target(harness):
    mean, sd = np.zeros(17), np.ones(17)
    _, st = harness(mean, sd, 600, 600, 1, 4)
    eps = st[..., 4]
    assert np.all(eps == eps[:, :1])  # frozen after tuning
    assert 0.3 < eps.mean() < 1.3  # 17-d unit normal: ~0.7 at 0.8 acceptance
    assert abs(st[..., 3].mean() - 0.8) < 0.08


def _correlated_target():
    """Pairs of variables that trade off almost exactly (like init / perm of an antigen in the abd posterior)."""
    rng = np.random.default_rng(9)
    cov = np.diag(np.exp(rng.uniform(np.log(0.05), np.log(2.0), size=17)) ** 2)
    for a, b in ((0, 1), (4, 5), (10, 12)):
        cov[a, b] = cov[b, a] = -0.995 * np.sqrt(cov[a, a] * cov[b, b])
    return rng.normal(size=17), cov


def test_dense_metric_shortens_trees_on_a_correlated_normal(harness):
    mean, cov = _correlated_target()
    prec = np.linalg.inv(cov)
    sd = np.sqrt(np.diag(cov))
    qd, sd_stats = harness(mean, sd, 1000, 1500, 21, 4, prec=prec, dense=True)
    q1, s1_stats = harness(mean, sd, 1000, 1500, 21, 4, prec=prec, dense=False)
    for q, st in ((qd, sd_stats), (q1, s1_stats)):
        flat = q.reshape(-1, 17)
        assert not st[..., 5].any()
        se = sd / np.sqrt(flat.shape[0] / 8)
        assert np.all(np.abs(flat.mean(0) - mean) < 5 * se), (flat.mean(0) - mean) / se
        emp = np.cov(flat.T)
        assert np.all(np.abs(np.sqrt(np.diag(emp)) / sd - 1) < 0.15)
        assert emp[0, 1] / np.sqrt(emp[0, 0] * emp[1, 1]) < -0.98  # the trade-off is there
    depth_dense, depth_diag = sd_stats[..., 1].mean(), s1_stats[..., 1].mean()
    assert depth_dense < 3.6 and depth_diag > depth_dense + 1.5, (depth_dense, depth_diag)
    assert 0.7 < sd_stats[..., 3].mean() < 0.95


def test_the_leapfrog_train_protocol_gives_the_classic_draws(harness, tmp_path):
    """The protocol between the host's tree logic and the device's per-chain state machine (abd_train.hpp; restated on the CPU in
    the harness): doubling directions drawn when the transition begins, the device taking leapfrog after leapfrog across halves
    and doublings, the host feeding on the records behind it and ignoring what the device took beyond the end of a tree.  Bit for
    bit the draws of the classic request / feed loop; with an evaluation of the start point first (the compound step's
    transition after a sweep) the same again."""
    lib = C.CDLL(harness.lib_path)
    dp = C.POINTER(C.c_double)
    lib.nuts_harness_run_trains.argtypes = [dp, dp, C.c_longlong, C.c_longlong, C.c_ulonglong, C.c_int, C.c_int, C.c_int, dp, dp]
    lib.nuts_harness_run_trains.restype = C.c_int
    mean, sd = _target()
    tune, draws, chains = 300, 200, 3
    q_ref, st_ref = harness(mean, sd, tune, draws, 7, chains)
    for eval_first, run_on in ((0, 0), (0, 3), (1, 0), (1, 5)):
        q = np.empty((chains, draws, 17))
        st = np.empty((chains, draws, 6))
        rc = lib.nuts_harness_run_trains(mean.ctypes.data_as(dp), sd.ctypes.data_as(dp), tune, draws, 7, chains, eval_first, run_on,
                                         q.ctypes.data_as(dp), st.ctypes.data_as(dp))
        assert rc == 0, (rc, eval_first, run_on)
        np.testing.assert_array_equal(q, q_ref, err_msg=f"eval_first={eval_first} run_on={run_on}")
        np.testing.assert_array_equal(st, st_ref)
    assert st_ref[..., 1].max() >= 3  # trees with several doublings, i.e. halves the device entered by itself
