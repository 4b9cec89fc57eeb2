"""
The native compound sampler (abd_sampler_*: NUTS + device Gibbs sweep, the chains as independent units) on the GPU.

NUTS draws cannot be compared with PyMC's (not importable offline; parity of the *sampler* is statistical,
parity of every logp / gradient / sweep it consumes is tested elsewhere).  What is pinned here:
 * bookkeeping: the recorded lp is the joint logp of the recorded theta at the resident discrete state;
   the device-side running means equal the mean of the per-draw Deterministics; the sweep inside the
   sampler is the same sweep abd_gibbs_sweep does (bit-exact state);
 * reproducibility for a seed; continuous-only mode leaves the discrete state alone;
 * statistics: with the discrete state frozen, the native sampler and an independent host NUTS agree on the
   posterior mean and spread of all 17 variables.
"""
import os

import numpy as np
import pytest

from abdpymc_amd import synthetic
from abdpymc_amd.data import TiterData
from oracle import abd_oracle as O
from tests.helpers import oracle_cohort_from_synth, random_sparse_cohort

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def test_td(golden_dir):
    return TiterData.from_disk(os.path.join(golden_dir, "test_cohort"))


def _start(m, chains, seed=0):
    pt = m.initial_point()
    q0 = np.empty((chains, 17))
    for c in range(chains):
        rng = np.random.default_rng([seed, c])
        m.ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
        q0[c] = m.ravel(pt) + 0.1 * rng.uniform(-1, 1, size=17)
    return q0


def test_bookkeeping_means_and_reproducibility(test_td):
    from abdpymc_amd.model import model

    m = model(test_td, splits=(14, 20), n_chains=3)
    ctx = m.ctx
    q0 = _start(m, 3)
    smp = ctx.sampler([0, 1, 2], q0, tune=40, seed=7, accumulate=True)
    th_t, st_t = smp.run(40)
    assert np.all(np.isfinite(st_t["lp"])) and st_t["tree_depth"].max() <= 10
    np.testing.assert_allclose(st_t["step_size"][:, 0], 0.25 / 17 ** 0.25, rtol=1e-14)  # PyMC's initial step
    assert len(np.unique(st_t["step_size"][0])) > 10  # and dual averaging moves it
    sums = [np.zeros((3,) + (test_td.n_gaps, test_td.n_inds)) for _ in range(3)]
    thetas = []
    for k in range(12):
        th, st = smp.run(1)
        thetas.append(th[:, 0])
        for c in range(3):
            # lp of the iteration = joint logp of its theta at the discrete state it left behind
            lp, _ = ctx.logp_dlogp(c, th[c, 0])
            assert abs(lp - st["lp"][c, 0]) <= 1e-11 * abs(lp)
            d_i, d_n, d_s = ctx.deterministics(c, th[c, 0])
            sums[0][c] += d_i
            sums[1][c] += d_n
            sums[2][c] += d_s
        assert st["gibbs_proposed"].min() > 0.6 * (26 * 10 + 10)  # transit_p = 0.8 of G*N + N dims
        assert np.all(st["step_size"] == st["step_size"][:, :1])
    for c in range(3):
        mi, mn, ms, n = smp.means(c)
        assert n == 12
        np.testing.assert_allclose(mi, sums[0][c] / 12, rtol=0, atol=1e-15)
        np.testing.assert_allclose(mn, sums[1][c] / 12, rtol=1e-13)
        np.testing.assert_allclose(ms, sums[2][c] / 12, rtol=1e-13)
    inv_mass, eps = smp.adaptation(1)
    assert inv_mass.shape == (17,) and np.all(inv_mass > 0) and eps == st["step_size"][1, 0]
    state = [ctx.get_discrete(c) for c in range(3)]
    smp.close()

    # same seed, same starting point: same draws and the same discrete state, bit for bit
    q0 = _start(m, 3)
    smp2 = ctx.sampler([0, 1, 2], q0, tune=40, seed=7, accumulate=False)
    th2, _ = smp2.run(52)
    np.testing.assert_array_equal(th2[:, :40], th_t)
    np.testing.assert_array_equal(th2[:, 40:], np.stack(thetas, axis=1))
    for c in range(3):
        i2, w2 = ctx.get_discrete(c)
        np.testing.assert_array_equal(i2, state[c][0])
        np.testing.assert_array_equal(w2, state[c][1])
    with pytest.raises(Exception):
        smp2.means(0)  # created without accumulate
    smp2.close()
    # another seed: other draws
    q0 = _start(m, 3)
    smp3 = ctx.sampler([0, 1, 2], q0, tune=40, seed=8)
    th3, _ = smp3.run(5)
    assert not np.array_equal(th3, th_t[:, :5])
    m.close()


def test_sweep_inside_the_sampler_is_abd_gibbs_sweep(test_td):
    from abdpymc_amd.model import model

    m = model(test_td, n_chains=2)
    ctx = m.ctx
    q0 = _start(m, 2)
    smp = ctx.sampler([0, 1], q0, tune=5, seed=3)
    th, st = smp.run(1)
    after = [ctx.get_discrete(c) for c in range(2)]
    # replay: same starting state, the sweep of iteration 0 at the theta the NUTS transition ended on
    _start(m, 2)
    acc, prop = ctx.gibbs_sweep([0, 1], th[:, 0], seed=(3 << 20) ^ 0x5EED, sweep=0)
    for c in range(2):
        i2, w2 = ctx.get_discrete(c)
        np.testing.assert_array_equal(i2, after[c][0])
        np.testing.assert_array_equal(w2, after[c][1])
    np.testing.assert_array_equal(acc, st["gibbs_accepted"][:, 0].astype(np.int64))
    np.testing.assert_array_equal(prop, st["gibbs_proposed"][:, 0].astype(np.int64))
    m.close()


def test_argument_errors(test_td):
    from abdpymc_amd.model import model

    m = model(test_td, n_chains=2)
    q0 = _start(m, 2)
    with pytest.raises(ValueError):
        m.ctx.sampler([0, 0], q0, tune=5)
    with pytest.raises(ValueError):
        m.ctx.sampler([0, 1], q0, tune=5, max_treedepth=40)
    with pytest.raises(ValueError):
        m.ctx.sampler([0, 1], q0, tune=5, target_accept=1.5)
    with pytest.raises(ValueError):
        m.ctx.sampler([0, 1], q0[:1], tune=5)
    bad = q0.copy()
    bad[1, 13] = 800.0  # sigma = e^800: logp is not finite
    with pytest.raises(ValueError):
        m.ctx.sampler([0, 1], bad, tune=5)
    m.close()


def test_frozen_discrete_state_matches_host_nuts():
    """Continuous part only (gibbs = 0) against the independent host NUTS of abdpymc_amd.sampler (slice variant)."""
    from abdpymc_amd.model import model
    from abdpymc_amd.sampler import DualAveraging, Nuts

    sc = synthetic.make_cohort(300, 40, seed=5)
    td = TiterData.from_arrays(40, 300, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)
    m = model(td, n_chains=4)
    ctx = m.ctx
    rng = np.random.default_rng(2)
    i_raw = (rng.random((40, 300)) < 1 / 40).astype(np.int8)
    w = (rng.random(300) < 0.5).astype(np.int8)
    q_init = synthetic.theta_init(40)
    for c in range(4):
        ctx.set_discrete(c, i_raw, w)
    q0 = q_init[None, :] + 0.05 * rng.uniform(-1, 1, size=(4, 17))
    smp = ctx.sampler([0, 1, 2, 3], q0, tune=500, seed=11, gibbs=False)
    smp.run(500)
    th, st = smp.run(500)
    i_after, w_after = ctx.get_discrete(2)
    np.testing.assert_array_equal(i_after, i_raw)
    np.testing.assert_array_equal(w_after, w)
    assert st["diverging"].sum() <= 5
    assert 0.6 < st["mean_tree_accept"].mean() < 0.95
    native = th.reshape(-1, 17)

    # host NUTS, one chain, same target
    def fn(x):
        return ctx.logp_dlogp(0, x)

    nuts = Nuts(fn, 17, np.random.default_rng(3))
    q = q0[0].copy()
    lp, g = fn(q)
    nuts.eps = nuts.find_reasonable_eps(q, lp, g)
    nuts.da = DualAveraging(nuts.eps)
    nuts.inv_mass = smp.adaptation(0)[0]  # the adapted metric; the host class adapts only the step size here
    host = []
    for it in range(1300):
        q, lp, g, _ = nuts.step(q, lp, g, adapt=it < 300)
        if it == 299:
            nuts.eps = nuts.da.final()
        if it >= 300:
            host.append(q.copy())
    host = np.asarray(host)
    sd = native.std(0)
    # means agree within 6 standard errors at a conservative effective sample size of n/5 for each run
    se = sd * np.sqrt(5.0 / native.shape[0] + 5.0 / host.shape[0])
    z = (native.mean(0) - host.mean(0)) / se
    assert np.abs(z).max() < 6, z
    np.testing.assert_allclose(host.std(0), sd, rtol=0.3)
    # chains agree with each other (split R-hat-like check on the native draws)
    per_chain = th.mean(1)
    assert np.all(np.abs(per_chain - native.mean(0)) < 6 * sd / np.sqrt(500 / 5)), (per_chain - native.mean(0)) / sd
    m.close()


def test_sample_entry_point_native_and_python_paths(test_td):
    from abdpymc_amd.model import model
    from abdpymc_amd.sampler import sample

    m = model(test_td, n_chains=2)
    res = sample(m, tune=20, draws=8, chains=2, seed=1)
    assert res["p"].shape == (2, 8) and res["i"].shape == (2, 8, 26, 10) and res["i_raw"].shape == (2, 8, 26, 10)
    assert res["mean_ab_s_mu"].shape == (2, 26, 10)
    np.testing.assert_allclose(res["mean_ab_n_mu"], res["ab_n_mu"].mean(1), rtol=1e-13)
    np.testing.assert_allclose(res["mean_i"], res["i"].mean(1), atol=1e-15)
    i_ref = O.constrain_infections(res["i_raw"][1, -1], np.asarray(test_td.pcrpos).T)
    np.testing.assert_array_equal(res["i"][1, -1], i_ref)
    lean = sample(m, tune=20, draws=8, chains=2, seed=1, record_deterministics=False, record_discrete=False)
    assert "i" not in lean and "i_raw" not in lean
    np.testing.assert_array_equal(lean["p"], res["p"])  # recording does not perturb the chain
    np.testing.assert_array_equal(lean["mean_ab_n_mu"], res["mean_ab_n_mu"])
    py = sample(m, tune=6, draws=3, chains=1, seed=1, native=False)
    assert py["i"].shape == (1, 3, 26, 10)
    m.close()


def test_cli_sharded_over_two_processes(tmp_path, golden_dir):
    """abdpymc-infer under torch.distributed.run, two ranks (both on this box's one GPU, gloo for the gather):
    each rank runs its chain with the random streams of its GLOBAL chain id, rank 0 writes all chains."""
    import subprocess
    import sys

    from abdpymc_amd.model import model
    from abdpymc_amd.sampler import sample

    import socket

    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "sharded"
    env = dict(os.environ, ABD_DIST_BACKEND="gloo", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "abdpymc_amd.cli", "--tune", "12", "--draws", "6", "--chains", "2", "--seed", "5",
           "--ititers_data", os.path.join(golden_dir, "test_cohort"), "--split_delta", "--device", "0", "--netcdf", str(out)]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    files = [f for f in tmp_path.iterdir()]
    assert len(files) == 1, files
    if files[0].suffix != ".npz":
        pytest.skip("ArviZ present: NetCDF written, draw comparison not implemented for it")
    z = np.load(files[0])
    assert z["p"].shape == (2, 6) and z["i"].shape == (2, 6, 26, 10)
    td = TiterData.from_disk(os.path.join(golden_dir, "test_cohort"))
    splits = td.calculate_splits(delta=True, omicron=False)
    for g in range(2):
        m = model(td, splits=splits, n_chains=1)
        one = sample(m, tune=12, draws=6, chains=1, seed=5, chain_offset=g)
        np.testing.assert_array_equal(one["p"][0], z["p"][g])
        np.testing.assert_array_equal(one["i_raw"][0], z["i_raw"][g])
        m.close()
    assert not np.array_equal(z["p"][0], z["p"][1])


def test_parameter_recovery_on_a_simulated_cohort():
    """End to end: ODs simulated from known dynamics through the model's own equations (synthetic.make_cohort),
    all-zero infections as the starting state; the compound sampler must find the infections and the 13
    identifiable parameters.  Exercises logp, every gradient entry, the sweep and both adaptations together."""
    from abdpymc_amd.model import model
    from abdpymc_amd.sampler import sample

    sc = synthetic.make_cohort(400, 40, seed=77)
    td = TiterData.from_arrays(40, 400, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)
    m = model(td, n_chains=2)
    res = sample(m, 400, 300, chains=2, seed=3, record_deterministics=False, record_discrete=False)
    assert res["stat_diverging"].sum() <= 3
    for name, truth in synthetic.TRUTH.items():
        d = res[name]
        assert abs(d.mean() - truth) < 5 * d.std() + 0.01 * abs(truth), (name, truth, d.mean(), d.std())
        assert d.std() < 0.1 * max(abs(truth), 1.0), (name, d.std())  # and the posterior is tight: the data identify it
    assert res["ab_s_p_waner"].mean() > 0.95  # everybody wanes in the simulation
    mean_i = res["mean_i"].mean(0)
    truth_i = sc.i_true.astype(bool)
    assert mean_i[truth_i].mean() > 0.8 and mean_i[~truth_i].mean() < 0.01
    assert abs(mean_i.sum() - truth_i.sum()) < 0.05 * truth_i.sum()
    m.close()


def test_dense_metric_on_the_model_posterior():
    """init and perm of an antigen trade off almost exactly in this posterior: with the full covariance as M^-1
    the trees are several times shorter and the answers the same."""
    from abdpymc_amd.model import model
    from abdpymc_amd.sampler import sample

    sc = synthetic.make_cohort(400, 40, seed=77)
    td = TiterData.from_arrays(40, 400, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos)
    m = model(td, n_chains=2)
    kw = dict(chains=2, seed=3, record_deterministics=False, record_discrete=False)
    diag = sample(m, 400, 300, **kw)
    dense = sample(m, 400, 300, dense_metric=True, **kw)
    assert dense["stat_n_steps"].mean() < 0.4 * diag["stat_n_steps"].mean(), (dense["stat_n_steps"].mean(), diag["stat_n_steps"].mean())
    assert dense["stat_diverging"].sum() <= 3
    for name, truth in synthetic.TRUTH.items():
        a, b = dense[name], diag[name]
        assert abs(a.mean() - truth) < 5 * a.std() + 0.01 * abs(truth), (name, truth, a.mean(), a.std())
        assert abs(a.mean() - b.mean()) < 1.0 * max(a.std(), b.std()), (name, a.mean(), b.mean())
        assert 0.6 < a.std() / b.std() < 1.6, (name, a.std(), b.std())
    # init + perm is pinned down far better than either: the correlation the diagonal metric cannot see
    s = dense["ab_n_init"] + dense["ab_n_perm"]
    assert s.std() < 0.3 * dense["ab_n_init"].std()
    m.close()


def test_more_chains_than_one_launch_holds(test_td):
    """20 chains: every lock-step leapfrog and every sweep is split into launches of <= 16 chains."""
    from abdpymc_amd.model import model

    m = model(test_td, n_chains=20)
    q0 = _start(m, 20)
    smp = m.ctx.sampler(np.arange(20), q0, tune=30, seed=2)
    th, st = smp.run(40)
    assert np.all(np.isfinite(st["lp"])) and th.shape == (20, 40, 17)
    for c in (0, 15, 16, 19):
        lp, _ = m.ctx.logp_dlogp(c, th[c, -1])
        assert abs(lp - st["lp"][c, -1]) <= 1e-11 * abs(lp)
    # all chains differ (own random streams, also across the two launch groups)
    assert len({tuple(np.round(th[c, -1], 12)) for c in range(20)}) == 20
    m.close()


def test_back_to_back_synchronous_calls_are_deterministic(test_td):
    """Regression: the NUTS driver issues synchronous evaluations ~17 us apart and polls their completion tag in
    mapped host memory.  That memory has to be COHERENT (fine-grained): with a plain mapped allocation the two
    64-byte halves of a result row could reach the host out of order -- tag visible, first half stale -- and about
    one run in fifty took a different trajectory.  Same seed, many runs: identical bits every time."""
    from abdpymc_amd.model import model

    m = model(test_td, n_chains=2)
    pt = m.initial_point()
    ref = None
    for r in range(120):
        q0 = np.empty((2, 17))
        for c in range(2):
            m.ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
            q0[c] = m.ravel(pt) + 0.1 * np.random.default_rng([1, c]).uniform(-1, 1, 17)
        smp = m.ctx.sampler([0, 1], q0, tune=20, seed=1)
        th, _ = smp.run(28)
        smp.close()
        if ref is None:
            ref = th
        else:
            assert np.array_equal(th, ref), r
    # ~100 k synchronous calls were waited for by polling their completion tags: none may have needed the
    # stream-synchronise fall-back (abd_wait_fallbacks)
    assert m.ctx.wait_fallbacks == 0
    m.close()


def test_run_record_equals_draw_by_draw_readback(test_td):
    """The batched recorder (device staging, block copies) against reading every draw back one at a time."""
    from abdpymc_amd.model import model

    m = model(test_td, splits=(14, 20), n_chains=2)
    ctx = m.ctx
    G, N = test_td.n_gaps, test_td.n_inds
    q0 = _start(m, 2)
    a = ctx.sampler([0, 1], q0, tune=10, seed=4)
    a.run(10)
    rec = dict(i_raw=np.full((2, 12, G, N), -1, np.int8), ab_s_waner=np.full((2, 12, N), -1, np.int8),
               i=np.full((2, 12, G, N), -1, np.int8), ab_n_mu=np.full((2, 12, G, N), np.nan), ab_s_mu=np.full((2, 12, G, N), np.nan))
    th1, _ = a.run_record(5, 0, **rec)
    assert np.all(rec["i_raw"][:, 5:] == -1) and np.all(np.isnan(rec["ab_n_mu"][:, 5:]))  # later draws untouched
    th2, _ = a.run_record(7, 5, **rec)
    a.close()
    th = np.concatenate([th1, th2], axis=1)
    # twin run, one iteration at a time
    q0 = _start(m, 2)
    b = ctx.sampler([0, 1], q0, tune=10, seed=4)
    b.run(10)
    for k in range(12):
        t, _ = b.run(1)
        np.testing.assert_array_equal(t[:, 0], th[:, k])
        for c in range(2):
            i_raw, w = ctx.get_discrete(c)
            d_i, d_n, d_s = ctx.deterministics(c, t[c, 0])
            np.testing.assert_array_equal(rec["i_raw"][c, k], i_raw)
            np.testing.assert_array_equal(rec["ab_s_waner"][c, k], w)
            np.testing.assert_array_equal(rec["i"][c, k], d_i)
            np.testing.assert_array_equal(rec["ab_n_mu"][c, k], d_n)
            np.testing.assert_array_equal(rec["ab_s_mu"][c, k], d_s)
    b.close()
    # only some variables; capacity checks
    c2 = ctx.sampler([0, 1], _start(m, 2), tune=0, seed=4)
    only = np.full((2, 3, G, N), np.nan)
    c2.run_record(3, 0, ab_s_mu=only)
    assert np.all(np.isfinite(only))
    with pytest.raises(ValueError):
        c2.run_record(3, 1, ab_s_mu=only)  # draws 1..3 do not fit capacity 3
    with pytest.raises(ValueError):
        c2.run_record(1, 0, i=np.zeros((2, 3, G, N)))  # wrong dtype
    m.close()


def test_a_chains_draws_do_not_depend_on_how_the_chains_are_grouped(test_td, monkeypatch):
    """The sampler runs its chains as independent units of 1-8 chains, each with its own launches in flight.  On the
    reference's cohorts (observation lists: a chain's sums never see the rest of a launch) a chain's draws -- continuous
    and discrete -- must be bit-identical whatever the unit size, and whether 2 or 5 chains run beside it."""
    from abdpymc_amd.model import model

    def run(n_chains, unit):
        monkeypatch.setenv("ABD_SAMPLER_UNIT", str(unit))
        m = model(test_td, splits=(14,), n_chains=n_chains)
        q0 = _start(m, n_chains, seed=3)
        smp = m.ctx.sampler(np.arange(n_chains), q0, tune=25, seed=9)
        th, st = smp.run(40)
        states = [m.ctx.get_discrete(c) for c in range(n_chains)]
        smp.close()
        m.close()
        return th, st, states

    def same(a, b, n_chains):
        np.testing.assert_array_equal(b[0], a[0][:n_chains])
        np.testing.assert_array_equal(b[1]["gibbs_accepted"], a[1]["gibbs_accepted"][:n_chains])
        for c in range(n_chains):
            np.testing.assert_array_equal(b[2][c][0], a[2][c][0])
            np.testing.assert_array_equal(b[2][c][1], a[2][c][1])

    # evaluations the host assembles (units of one chain would run leapfrog trains, whose closed forms are the device's:
    # equal to rounding, not bit for bit -- test_observation_list_trains_follow_the_host_driven_chain)
    monkeypatch.setenv("ABD_SAMPLER_TRAINS", "0")
    ref = run(5, 1)
    for n_chains, unit in ((5, 2), (5, 4), (5, 8), (2, 1), (2, 2)):
        same(ref, run(n_chains, unit), n_chains)
    assert len({tuple(ref[0][c, -1]) for c in range(5)}) == 5
    # leapfrog trains: a chain's draws do not depend on how many chains run beside it
    monkeypatch.setenv("ABD_SAMPLER_TRAINS", "1")
    ref_t = run(5, 1)
    same(ref_t, run(2, 1), 2)
    same(ref_t, run(5, 1), 5)


def test_host_threads_do_not_change_a_draw(test_td, monkeypatch):
    """The units of an observation-list cohort are driven by up to four host threads (thread t the units t, t + 4, ...).
    Whatever the number of threads and however the 6 chains are split into units, every chain's draws, statistics,
    recorded Deterministics and final discrete state are the single-thread ones bit for bit; a dense cohort driven by
    several threads (not the default there) is as reproducible as on one."""
    from abdpymc_amd._native import Context
    from abdpymc_amd.model import model

    G, N, C = test_td.n_gaps, test_td.n_inds, 6

    def run(threads, unit):
        monkeypatch.setenv("ABD_SAMPLER_THREADS", str(threads))
        monkeypatch.setenv("ABD_SAMPLER_UNIT", str(unit))
        m = model(test_td, splits=(14,), n_chains=C)
        smp = m.ctx.sampler(np.arange(C), _start(m, C, seed=5), tune=20, seed=11, accumulate=True)
        rec = dict(i=np.zeros((C, 30, G, N), np.int8), ab_s_mu=np.zeros((C, 30, G, N)))
        th, st = smp.run_record(30, 0, **rec)
        means = [smp.means(c)[1] for c in range(C)]
        states = [m.ctx.get_discrete(c) for c in range(C)]
        fb = m.ctx.wait_fallbacks
        smp.close()
        m.close()
        return th, st, rec, means, states, fb

    def check(ref, cases):
        for threads, unit in cases:
            compare(ref, run(threads, unit))

    def compare(ref, got):
        np.testing.assert_array_equal(got[0], ref[0])
        for k in ref[1]:
            if k != "t_done":  # (host clock)
                np.testing.assert_array_equal(got[1][k], ref[1][k], err_msg=k)
        np.testing.assert_array_equal(got[2]["i"], ref[2]["i"])
        np.testing.assert_array_equal(got[2]["ab_s_mu"], ref[2]["ab_s_mu"])
        for c in range(C):
            np.testing.assert_array_equal(got[3][c], ref[3][c])
            np.testing.assert_array_equal(got[4][c][0], ref[4][c][0])
            np.testing.assert_array_equal(got[4][c][1], ref[4][c][1])
        assert got[5] == 0

    # evaluations the host assembles: any number of threads, any split into units
    monkeypatch.setenv("ABD_SAMPLER_TRAINS", "0")
    check(run(1, 2), ((4, 1), (3, 2), (2, 1), (16, 1), (4, 8)))
    # leapfrog trains (units of one chain): any number of threads
    monkeypatch.setenv("ABD_SAMPLER_TRAINS", "1")
    check(run(1, 1), ((4, 1), (2, 1), (16, 1)))
    monkeypatch.delenv("ABD_SAMPLER_TRAINS")

    sc = synthetic.make_cohort(300, 40, seed=4)

    def run_dense(threads):
        monkeypatch.setenv("ABD_SAMPLER_THREADS", str(threads))
        monkeypatch.setenv("ABD_SAMPLER_UNIT", "1")
        ctx = Context(40, 300, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=(17,), n_chains=4)
        for c in range(4):
            ctx.set_discrete(c, *synthetic.make_chain_state(300, 40, c))
        smp = ctx.sampler(np.arange(4), np.stack([synthetic.make_thetas(40, 1, c)[0] for c in range(4)]), tune=10, seed=2)
        th, st = smp.run(14)
        smp.close()
        ctx.close()
        return th, st

    th1, st1 = run_dense(1)
    th4, st4 = run_dense(4)
    np.testing.assert_array_equal(th1, th4)
    np.testing.assert_array_equal(st1["gibbs_accepted"], st4["gibbs_accepted"])


def test_dense_unit_launch_that_sums_its_own_partials_gives_the_same_draws(monkeypatch):
    """A sampler unit's dense launch sums its own partial rows (last workgroup in, device-coherent re-read; the default)
    or leaves that to a second launch (ABD_DENSE_OWN_SUM=0): same order of additions, so the same draws bit for bit,
    with units of one and of two chains.  (Leapfrog trains off: they exist only with the own sum and assemble logp on the
    device -- test_leapfrog_trains_follow_the_host_driven_chain.)"""
    from abdpymc_amd._native import Context

    sc = synthetic.make_cohort(700, 130, seed=6)
    monkeypatch.setenv("ABD_SAMPLER_TRAINS", "0")

    def run(own, unit):
        monkeypatch.setenv("ABD_DENSE_OWN_SUM", own)
        monkeypatch.setenv("ABD_SAMPLER_UNIT", unit)
        ctx = Context(130, 700, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=(40, 90), n_chains=4)
        for c in range(4):
            ctx.set_discrete(c, *synthetic.make_chain_state(700, 130, c))
        smp = ctx.sampler(np.arange(4), np.stack([synthetic.make_thetas(130, 1, c)[0] for c in range(4)]), tune=10, seed=8)
        th, st = smp.run(16)
        states = [ctx.get_discrete(c) for c in range(4)]
        fb = ctx.wait_fallbacks
        smp.close()
        ctx.close()
        return th, st, states, fb

    for unit in ("1", "2"):
        th0, st0, s0, _ = run("0", unit)
        th1, st1, s1, fb = run("1", unit)
        assert fb == 0
        np.testing.assert_array_equal(th0, th1)
        np.testing.assert_array_equal(st0["lp"], st1["lp"])
        for (a, b), (c, d) in zip(s0, s1):
            np.testing.assert_array_equal(a, c)
            np.testing.assert_array_equal(b, d)


def test_units_on_a_dense_cohort_are_deterministic_and_record_like_a_twin_run():
    """Dense cohort (the dense evaluation kernel and the lane-per-proposal sweep), 3 chains as 3 units: two runs give the
    same bits; recording while running does not perturb the chains; every recorded `i` is the constrained `i_raw` of
    the same draw and the recorded lp is the joint logp there."""
    from abdpymc_amd._native import Context

    sc = synthetic.make_cohort(300, 40, seed=4)
    coh = oracle_cohort_from_synth(sc)
    G, N, C = 40, 300, 3

    def run(record):
        ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=(17,), n_chains=C)
        q0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
        for c in range(C):
            ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
        smp = ctx.sampler(np.arange(C), q0, tune=10, seed=2)
        rec = {}
        if record:
            rec = dict(i_raw=np.zeros((C, 14, G, N), np.int8), i=np.zeros((C, 14, G, N), np.int8), ab_s_waner=np.zeros((C, 14, N), np.int8))
            th, st = smp.run_record(14, 0, **rec)
        else:
            th, st = smp.run(14)
        smp.close()
        ctx.close()
        return th, st, rec

    th_a, st_a, _ = run(False)
    th_b, st_b, rec = run(True)
    th_c, _, _ = run(False)
    np.testing.assert_array_equal(th_a, th_c)
    np.testing.assert_array_equal(th_a, th_b)
    np.testing.assert_array_equal(st_a["lp"], st_b["lp"])
    for c in range(C):
        for k in (0, 7, 13):
            i_ref = O.constrain_infections(rec["i_raw"][c, k], np.asarray(sc.pcrpos).T, (17,))
            np.testing.assert_array_equal(rec["i"][c, k], i_ref)
            lp = O.logp_dlogp(th_b[c, k], rec["i_raw"][c, k], rec["ab_s_waner"][c, k], coh, (17,))[0]
            assert abs(lp - st_b["lp"][c, k]) <= 1e-9 * abs(lp)


def test_observation_list_trains_follow_the_host_driven_chain(test_td, monkeypatch):
    """Leapfrog trains on the reference's cohorts (observation lists, units of one chain, the units on their own host threads):
    the lane-per-observation kernel of a train launch sums its own partial rows and its last workgroup does the leapfrog.
    Same trees and the same points to rounding as the host-driven chain over the first transitions, exactly repeatable, and
    no completion wait falls back to a stream synchronise."""
    from abdpymc_amd.model import model

    C = 4

    def run(trains):
        monkeypatch.setenv("ABD_SAMPLER_TRAINS", trains)
        monkeypatch.setenv("ABD_SAMPLER_UNIT", "1")
        m = model(test_td, splits=(14,), n_chains=C)
        smp = m.ctx.sampler(np.arange(C), _start(m, C, seed=4), tune=30, seed=7)
        th, st = smp.run(40)
        states = [m.ctx.get_discrete(c) for c in range(C)]
        fb = m.ctx.wait_fallbacks
        smp.close()
        m.close()
        return th, st, states, fb

    t_host, s_host, _, _ = run("0")
    t_a, s_a, d_a, fb_a = run("1")
    t_b, s_b, d_b, fb_b = run("1")
    assert fb_a == 0 and fb_b == 0
    assert np.array_equal(t_a, t_b) and all(np.array_equal(s_a[k], s_b[k]) for k in s_a if k != "t_done")
    assert all(np.array_equal(d_a[c][0], d_b[c][0]) and np.array_equal(d_a[c][1], d_b[c][1]) for c in range(C))
    assert np.isfinite(t_a).all() and (s_a["n_steps"] >= 1).all() and s_a["n_steps"].max() > 3
    assert np.array_equal(s_a["n_steps"][:, :3], s_host["n_steps"][:, :3])
    np.testing.assert_allclose(t_a[:, :3], t_host[:, :3], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(s_a["lp"][:, :3], s_host["lp"][:, :3], rtol=1e-9)


@pytest.mark.parametrize("unit,C", [("1", 3), ("2", 4), ("4", 4)])
def test_leapfrog_trains_follow_the_host_driven_chain(monkeypatch, unit, C):
    """Leapfrog trains (dense cohort; units of one, two and four chains; on by default): the launch that evaluates a point
    assembles logp and gradient on the device, finishes the leapfrog -- across the halves and doublings of the tree, from the
    chain's state machine in device memory -- and leaves the next point for the launch queued behind it; the host runs the
    tree logic on the records.  Same transition, same random stream as the host-driven chain (same unit, so the same range
    split); the device's closed forms (exp, log1p) round differently from the host's, so the chains agree to rounding, not
    bit for bit: the first transitions build the same trees and end within 1e-7 of each other; and a train run repeats
    itself exactly, whatever the look-ahead did (it depends on timing only)."""
    from abdpymc_amd._native import Context

    N, G = 700, 70
    sc = synthetic.make_cohort(N, G, seed=9)
    monkeypatch.setenv("ABD_SAMPLER_UNIT", unit)

    def run(trains):
        monkeypatch.setenv("ABD_SAMPLER_TRAINS", trains)
        ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
        for c in range(C):
            ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
        th0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
        smp = ctx.sampler(np.arange(C), th0, tune=50, seed=5, gibbs=True)
        theta, st = smp.run(12)
        fb = ctx.wait_fallbacks
        smp.close()
        ctx.close()
        return theta, st, fb

    t_host, s_host, _ = run("0")
    t_a, s_a, fb_a = run("1")
    t_b, s_b, fb_b = run("1")
    assert fb_a == 0 and fb_b == 0
    assert np.array_equal(t_a, t_b) and all(np.array_equal(s_a[k], s_b[k]) for k in s_a if k != "t_done")
    assert np.isfinite(t_a).all() and (s_a["n_steps"] >= 1).all()
    # the first transitions: same trees, same points to rounding (later ones may part ways: a decision on a knife's edge)
    assert np.array_equal(s_a["n_steps"][:, :3], s_host["n_steps"][:, :3])
    np.testing.assert_allclose(t_a[:, :3], t_host[:, :3], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(s_a["lp"][:, :3], s_host["lp"][:, :3], rtol=1e-9)


def test_observation_list_trains_beyond_256_gaps_report_the_oracles_logp(monkeypatch):
    """Leapfrog trains through the 8-word form of the lane-per-observation kernel (n_gaps = 300, two splits): every recorded
    draw's `i_raw` / `ab_s_waner` and theta give, in the oracle, the joint logp the sampler reported for that draw (the
    value its device-side epilogue assembled after the sweep), and the run repeats itself bit for bit."""
    from abdpymc_amd._native import Context

    G, N, C, n_it = 300, 29, 3, 12
    splits = (100, 201)
    coh = random_sparse_cohort(N, G, 1500, 1200, seed=77)
    monkeypatch.setenv("ABD_SAMPLER_UNIT", "1")
    monkeypatch.setenv("ABD_SAMPLER_TRAINS", "1")

    def run():
        ctx = Context(G, N, (coh.s.idx_gap, coh.s.idx_ind, coh.s.log_dilution, coh.s.od),
                      (coh.n.idx_gap, coh.n.idx_ind, coh.n.log_dilution, coh.n.od), coh.vacs, coh.pcrpos, splits=splits, n_chains=C)
        assert not ctx.is_dense
        rng = np.random.default_rng(5)
        q0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
        for c in range(C):
            ctx.set_discrete(c, (rng.random((G, N)) < 2.0 / G).astype(np.int8), (rng.random(N) < 0.5).astype(np.int8))
        smp = ctx.sampler(np.arange(C), q0, tune=8, seed=21)
        rec = dict(i_raw=np.zeros((C, n_it, G, N), np.int8), ab_s_waner=np.zeros((C, n_it, N), np.int8))
        th, st = smp.run_record(n_it, 0, **rec)
        fb = ctx.wait_fallbacks
        smp.close()
        ctx.close()
        return th, st, rec, fb

    th, st, rec, fb = run()
    th2, st2, rec2, fb2 = run()
    assert fb == 0 and fb2 == 0
    assert np.array_equal(th, th2) and np.array_equal(st["lp"], st2["lp"]) and np.array_equal(rec["i_raw"], rec2["i_raw"])
    assert (st["n_steps"] >= 1).all() and st["n_steps"].max() > 3 and np.isfinite(th).all()
    for c in range(C):
        for k in (0, 5, n_it - 1):
            lp = O.logp_dlogp(th[c, k], rec["i_raw"][c, k], rec["ab_s_waner"][c, k], coh, splits)[0]
            assert abs(lp - st["lp"][c, k]) <= 1e-9 * abs(lp), (c, k, lp, st["lp"][c, k])


def test_thinned_recording_is_every_kth_draw_of_the_unthinned_run(test_td):
    """--thin K (the reference thins afterwards, subsample_idata.py): recording every K-th draw does not perturb
    the chains, the per-draw (gap, ind) arrays are bit for bit draws 0, K, 2K, ... of the unthinned run -- whatever the
    chunking of the calls -- and the 17 scalars, the statistics and the posterior means still cover every draw."""
    from abdpymc_amd import sampler
    from abdpymc_amd.model import model

    def run(thin, chunk):
        m = model(test_td, splits=(14,), n_chains=3)
        res = sampler.sample_native(m, 12, 23, chains=3, seed=4, thin=thin, chunk=chunk)
        m.close()
        return res

    full = run(1, 50)
    for thin, chunk in ((5, 50), (5, 7), (4, 9), (23, 50), (40, 50)):
        got = run(thin, chunk)
        idx = np.arange(0, 23, thin)
        assert got["draw_index"].tolist() == [idx.tolist()] * 3
        for name in ("i_raw", "ab_s_waner", "i", "ab_n_mu", "ab_s_mu"):
            assert got[name].shape[1] == len(idx)
            np.testing.assert_array_equal(got[name], full[name][:, idx], err_msg=f"{name} thin={thin} chunk={chunk}")
        for name in ("p", "ab_n_rho", "stat_lp", "stat_n_steps", "mean_i", "mean_ab_n_mu", "mean_ab_s_mu"):
            np.testing.assert_array_equal(got[name], full[name], err_msg=name)
    with pytest.raises(ValueError, match="budget"):
        m = model(test_td, splits=(14,), n_chains=3)
        try:
            sampler.sample_native(m, 2, 23, chains=3, budget_bytes=1000)
        finally:
            m.close()


def test_cli_on_a_config3_sized_cohort_directory_stays_within_the_host_budget(tmp_path):
    """abdpymc-infer at BASELINE sizes (missing 3): a 10 000 x 200 cohort DIRECTORY in the reference's format,
    4 chains, --thin so that 2 draws per chain are kept.  The run must finish with a bounded host footprint (the default
    recording of every draw is refused with the thin that fits) and write the posterior means and the thinned draws."""
    import resource
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_cohort_dir import write_cohort_dir

    d = tmp_path / "cohort"
    write_cohort_dir(str(d), 10000, 200)
    out = tmp_path / "post"
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "abdpymc_amd.cli", "--ititers_data", str(d), "--chains", "4", "--netcdf", str(out)]
    # (1) unthinned at this size: refused before sampling, with the thin that fits
    r = subprocess.run(base + ["--tune", "5", "--draws", "200"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "thin >=" in r.stderr, r.stderr[-2000:]
    # (2) thinned: runs, bounded memory
    r = subprocess.run(base + ["--tune", "12", "--draws", "20", "--thin", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rss_gb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 2 ** 20  # (KiB -> GiB; the largest child so far)
    assert rss_gb < 16.0, rss_gb
    z = np.load(str(out) + ".npz")
    assert z["mean_i"].shape == (4, 200, 10000) and z["mean_ab_n_mu"].shape == (4, 200, 10000)
    assert z["i"].shape == (4, 2, 200, 10000) and z["draw_index"].tolist() == [[0, 10]] * 4
    assert z["p"].shape == (4, 20) and np.isfinite(z["stat_lp"]).all()
    assert np.isfinite(z["mean_ab_s_mu"]).all() and 0.0 <= z["mean_i"].min() and z["mean_i"].max() <= 1.0


def test_set_adaptation_installs_step_size_and_metric(test_td):
    """abd_sampler_set_adaptation (pooled adaptation between runs): the chain takes the step size and the diagonal metric it is
    given -- a draw run afterwards reports that step size -- and bad values are refused."""
    from abdpymc_amd.model import model

    m = model(test_td, splits=(14,), n_chains=2)
    smp = m.ctx.sampler(np.arange(2), _start(m, 2, seed=2), tune=15, seed=1, gibbs=False)
    smp.run(15)
    im0, eps0 = smp.adaptation(0)
    im1, eps1 = smp.adaptation(1)
    im = np.sqrt(im0 * im1)
    eps = float(np.sqrt(eps0 * eps1))
    for k in range(2):
        smp.set_adaptation(k, im, eps)
        got_im, got_eps = smp.adaptation(k)
        np.testing.assert_array_equal(got_im, im)
        assert got_eps == eps
    _, st = smp.run(5)
    assert np.all(st["step_size"] == eps)
    smp.set_adaptation(0, None, 0.5 * eps)  # step size alone
    assert smp.adaptation(0)[1] == 0.5 * eps and np.array_equal(smp.adaptation(0)[0], im)
    with pytest.raises(ValueError):
        smp.set_adaptation(0, np.zeros(17), eps)
    with pytest.raises(ValueError):
        smp.set_adaptation(5, im, eps)
    smp.close()
    m.close()


def test_train_units_of_several_chains_are_repeatable(monkeypatch):
    """Dense cohort large enough for the units' full launch shape (a grid that fills the chip), chains in train units of 2 and
    of 4: a chain's sweeps run on its own stream beside its unit's launches for the other chains, so which chains step in
    which launch depends on timing -- a chain's draws must not (the launch shape, and with it the order of the sums, is the
    unit's, whatever the other chains have pending).  Twice the same bits, no completion wait falling back."""
    from abdpymc_amd._native import Context

    N, G, C = 1500, 200, 4
    sc = synthetic.make_cohort(N, G, seed=12)

    def run(unit):
        monkeypatch.setenv("ABD_SAMPLER_UNIT", unit)
        ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=C)
        for c in range(C):
            ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
        th0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
        smp = ctx.sampler(np.arange(C), th0, tune=10, seed=3)
        th, st = smp.run(16)
        states = [ctx.get_discrete(c) for c in range(C)]
        fb = ctx.wait_fallbacks
        smp.close()
        ctx.close()
        return th, st, states, fb

    for unit in ("2", "4"):
        a, b = run(unit), run(unit)
        assert a[3] == 0 and b[3] == 0
        np.testing.assert_array_equal(a[0], b[0])
        for k in a[1]:
            if k != "t_done":
                np.testing.assert_array_equal(a[1][k], b[1][k], err_msg=k)
        for (i0, w0), (i1, w1) in zip(a[2], b[2]):
            np.testing.assert_array_equal(i0, i1)
            np.testing.assert_array_equal(w0, w1)
        assert np.isfinite(a[0]).all() and (a[1]["n_steps"] >= 1).all() and a[1]["gibbs_proposed"].min() > 0
