"""
GPU parity: the HIP path (through the C ABI, via ctypes) against the CPU oracle on the same seeded
inputs.  Tolerance: 1e-6 relative on fp64 logp and on every gradient entry (north_star); in practice
the agreement is ~1e-12.  Gradient entries are compared relative to max(|g_k|, 1e-6 * max|g|): an
entry that cancels to ~0 has no meaningful relative error.
"""
import numpy as np
import pytest

from abdpymc_amd import synthetic
from oracle import abd_oracle as O
from tests.helpers import oracle_cohort_from_synth, random_sparse_cohort

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def _ctx(coh, splits=None, ignore=False, n_chains=1, storage="f64"):
    from abdpymc_amd._native import Context

    return Context(
        coh.n_gaps,
        coh.n_inds,
        (coh.s.idx_gap, coh.s.idx_ind, coh.s.log_dilution, coh.s.od),
        (coh.n.idx_gap, coh.n.idx_ind, coh.n.log_dilution, coh.n.od),
        coh.vacs,
        None if ignore else coh.pcrpos,
        splits=splits,
        n_chains=n_chains,
        storage=storage,
    )


def _state(coh, seed, rate=None):
    rng = np.random.default_rng(seed)
    rate = rate if rate is not None else 2.0 / coh.n_gaps
    i_raw = (rng.random((coh.n_gaps, coh.n_inds)) < rate).astype(np.int8)
    w = (rng.random(coh.n_inds) < 0.5).astype(np.int8)
    theta = synthetic.theta_init(coh.n_gaps) + 0.3 * rng.standard_normal(17)
    return theta, i_raw, w


def assert_close(lp, g, lp_ref, g_ref, rtol=RTOL):
    assert abs(lp - lp_ref) <= rtol * abs(lp_ref), (lp, lp_ref)
    scale = np.maximum(np.abs(g_ref), 1e-6 * np.abs(g_ref).max())
    err = np.abs(g - g_ref) / scale
    assert err.max() <= rtol, (err, g, g_ref)


@pytest.mark.parametrize("G,N", [(20, 23), (60, 100), (64, 65), (65, 7), (200, 50), (256, 9), (2, 5), (257, 70), (300, 130), (512, 67)])
@pytest.mark.parametrize("splits", [None, "one", "two"])
def test_dense_parity(G, N, splits):
    coh = oracle_cohort_from_synth(synthetic.make_cohort(N, G, seed=G * 1000 + N))
    sp = {None: None, "one": (G // 2,), "two": (G // 3, (2 * G) // 3)}[splits]
    ctx = _ctx(coh, sp)
    assert ctx.is_dense
    theta, i_raw, w = _state(coh, 11)
    ctx.set_discrete(0, i_raw, w)
    lp, g = ctx.logp_dlogp(0, theta)
    lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh, sp)
    assert_close(lp, g, lp_ref, g_ref)
    assert abs(ctx.logp(0, theta) - lp_ref) <= RTOL * abs(lp_ref)
    ctx.close()


def test_dense_matches_literal_dense_oracle():
    """Against the reference-faithful (G,G,N) formulation (abd.py:242-274), small enough to fit."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(40, 31, seed=77))
    ctx = _ctx(coh, (14, 20))
    theta, i_raw, w = _state(coh, 5)
    ctx.set_discrete(0, i_raw, w)
    lp = ctx.logp(0, theta)
    ref = O.joint_logp(theta, i_raw, w, coh, (14, 20), dense=True)
    assert abs(lp - ref) <= RTOL * abs(ref)


# the two kernels for sparse observation lists: lane per observation (abd_obs.hpp, the default below 128
# observations per individual and antigen) and wave per individual; ABD_OBS_LANES forces one (read at create)
@pytest.mark.parametrize("lanes", ["1", "0"])
@pytest.mark.parametrize("ignore", [False, True])
@pytest.mark.parametrize("splits", [None, (10,), (0,), (26,), (8, 18)])
def test_sparse_parity(ignore, splits, lanes, monkeypatch):
    monkeypatch.setenv("ABD_OBS_LANES", lanes)
    coh = random_sparse_cohort(37, 26, 900, 700, seed=9)
    ctx = _ctx(coh, splits, ignore)
    assert not ctx.is_dense
    theta, i_raw, w = _state(coh, 12)
    ctx.set_discrete(0, i_raw, w)
    lp, g = ctx.logp_dlogp(0, theta)
    lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh, splits, ignore)
    assert_close(lp, g, lp_ref, g_ref)
    assert abs(ctx.logp(0, theta) - lp_ref) <= RTOL * abs(lp_ref)  # the logp-only instantiation


@pytest.mark.parametrize("lanes", [None, "1", "0"])
def test_sparse_many_obs_per_individual(lanes, monkeypatch):
    """More than 64 observations per individual: several lane chunks per wave (wave per individual, which is
    what the library picks by itself here) / individuals spanning several wavefronts (lane per observation)."""
    if lanes is not None:
        monkeypatch.setenv("ABD_OBS_LANES", lanes)
    coh = random_sparse_cohort(5, 40, 1500, 1100, seed=10)
    ctx = _ctx(coh, (13, 30), n_chains=3)
    thetas, refs = [], []
    for c in range(3):
        theta, i_raw, w = _state(coh, 13 + c)
        ctx.set_discrete(c, i_raw, w)
        thetas.append(theta)
        refs.append(O.logp_dlogp(theta, i_raw, w, coh, (13, 30)))
    lp, g = ctx.logp_dlogp_batch([0, 1, 2], np.array(thetas))
    for c in range(3):
        assert_close(lp[c], g[c], *refs[c])


def test_sparse_lists_longer_than_the_grid(monkeypatch):
    """The lane-per-observation grid is capped (8 workgroups per CU and list): lanes stride over longer lists."""
    monkeypatch.setenv("ABD_OBS_LANES", "1")
    coh = random_sparse_cohort(3000, 200, 700_000, 600_000, seed=11)
    ctx = _ctx(coh, (70, 140), n_chains=2)
    refs, thetas = [], []
    for c in range(2):
        theta, i_raw, w = _state(coh, 40 + c)
        ctx.set_discrete(c, i_raw, w)
        thetas.append(theta)
        refs.append(O.logp_dlogp(theta, i_raw, w, coh, (70, 140)))
    lp, g = ctx.logp_dlogp_batch([0, 1], np.array(thetas))
    for c in range(2):
        assert_close(lp[c], g[c], *refs[c])


def test_empty_antigen_and_individuals_without_obs():
    coh = random_sparse_cohort(9, 12, 0, 30, seed=2)
    ctx = _ctx(coh)
    theta, i_raw, w = _state(coh, 14)
    ctx.set_discrete(0, i_raw, w)
    lp, g = ctx.logp_dlogp(0, theta)
    lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh)
    assert_close(lp, g, lp_ref, g_ref)


@pytest.mark.parametrize("n_chains", [1, 2, 3, 4, 5, 8, 17])
def test_batched_chains(n_chains):
    """Chains share the OD panels in one launch (1, 2 or 4 per wave); each must equal its own eval."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(300, 70, seed=21))
    sp = (30,)
    ctx = _ctx(coh, sp, n_chains=n_chains)
    thetas, refs = [], []
    for c in range(n_chains):
        theta, i_raw, w = _state(coh, 100 + c)
        ctx.set_discrete(c, i_raw, w)
        thetas.append(theta)
        refs.append(O.logp_dlogp(theta, i_raw, w, coh, sp))
    lp, g = ctx.logp_dlogp_batch(np.arange(n_chains), np.array(thetas))
    for c in range(n_chains):
        assert_close(lp[c], g[c], *refs[c])
    # permuted chain order + async path
    perm = np.arange(n_chains)[::-1].copy()
    ctx.enqueue(3, perm, np.array(thetas)[perm])
    ctx.wait()
    lp2, g2 = ctx.fetch(3, n_chains)
    # a chain may land in a differently shaped launch group (4, 2 or 1 chains per workgroup) when the
    # order changes, which changes the summation order: equal to rounding, not bitwise
    np.testing.assert_allclose(lp2, lp[perm], rtol=1e-13)
    np.testing.assert_allclose(g2, g[perm], rtol=1e-10, atol=1e-10 * np.abs(g).max())


def test_bitwise_reproducible():
    coh = oracle_cohort_from_synth(synthetic.make_cohort(500, 100, seed=22))
    ctx = _ctx(coh)
    theta, i_raw, w = _state(coh, 15)
    ctx.set_discrete(0, i_raw, w)
    a = ctx.logp_dlogp(0, theta)
    for _ in range(5):
        b = ctx.logp_dlogp(0, theta)
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1], b[1])


def test_heavy_infection_masks():
    """Dense i_raw (all ones / alternating) exercises the greedy 3-gap recurrence and first-in-chunk."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(70, 130, seed=23))
    for sp in (None, (50,), (40, 90)):
        ctx = _ctx(coh, sp)
        for rate in (1.0, 0.5, 0.0):
            theta, i_raw, w = _state(coh, 16, rate=rate)
            ctx.set_discrete(0, i_raw, w)
            lp, g = ctx.logp_dlogp(0, theta)
            lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh, sp)
            assert_close(lp, g, lp_ref, g_ref)
            i, mun, mus = ctx.deterministics(0, theta)
            i_ref, mun_ref, mus_ref = O.deterministics(theta, i_raw, w, coh, sp)
            np.testing.assert_array_equal(i, i_ref)  # integer work: bit-exact
            np.testing.assert_allclose(mun, mun_ref, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(mus, mus_ref, rtol=1e-12, atol=1e-12)
        ctx.close()


def test_fp32_storage():
    """fp32-held panels, fp64 arithmetic: equals the oracle run on the fp32-rounded panels."""
    sc = synthetic.make_cohort(200, 64, seed=24)
    coh = oracle_cohort_from_synth(sc)
    ctx = _ctx(coh, storage="f32")
    theta, i_raw, w = _state(coh, 17)
    ctx.set_discrete(0, i_raw, w)
    lp, g = ctx.logp_dlogp(0, theta)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    coh32 = O.Cohort(
        coh.n_gaps, coh.n_inds, coh.vacs, coh.pcrpos,
        O.AntigenObs(coh.s.idx_gap, coh.s.idx_ind, r32(coh.s.log_dilution), r32(coh.s.od)),
        O.AntigenObs(coh.n.idx_gap, coh.n.idx_ind, r32(coh.n.log_dilution), r32(coh.n.od)),
    )
    lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh32)
    assert_close(lp, g, lp_ref, g_ref)
    # and within fp32 rounding of the fp64 panels: tolerance 1e-4 relative, stated
    lp64, g64 = O.logp_dlogp(theta, i_raw, w, coh)
    assert abs(lp - lp64) <= 1e-4 * abs(lp64)


@pytest.mark.parametrize("lanes", ["1", "0"])
def test_fp32_storage_sparse_lists(lanes, monkeypatch):
    """fp32-held observation lists through both sparse kernels (logp + gradient and the logp-only form)."""
    monkeypatch.setenv("ABD_OBS_LANES", lanes)
    coh = random_sparse_cohort(50, 40, 1500, 1300, seed=3)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    coh32 = O.Cohort(
        coh.n_gaps, coh.n_inds, coh.vacs, coh.pcrpos,
        O.AntigenObs(coh.s.idx_gap, coh.s.idx_ind, r32(coh.s.log_dilution), r32(coh.s.od)),
        O.AntigenObs(coh.n.idx_gap, coh.n.idx_ind, r32(coh.n.log_dilution), r32(coh.n.od)),
    )
    ctx = _ctx(coh, (13, 27), n_chains=2, storage="f32")
    assert not ctx.is_dense
    for c in range(2):
        theta, i_raw, w = _state(coh, 60 + c)
        ctx.set_discrete(c, i_raw, w)
        lp, g = ctx.logp_dlogp(c, theta)
        lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh32, (13, 27))
        assert_close(lp, g, lp_ref, g_ref)
        assert abs(ctx.logp(c, theta) - lp_ref) <= RTOL * abs(lp_ref)


def test_flip_discrete_matches_reupload():
    coh = oracle_cohort_from_synth(synthetic.make_cohort(33, 40, seed=25))
    ctx = _ctx(coh, (20,))
    theta, i_raw, w = _state(coh, 18)
    ctx.set_discrete(0, i_raw, w)
    G, N = coh.n_gaps, coh.n_inds
    rng = np.random.default_rng(3)
    for _ in range(10):
        flat = int(rng.integers(0, G * N + N))
        ctx.flip_discrete(0, flat)
        if flat < G * N:
            i_raw.ravel()[flat] ^= 1
        else:
            w[flat - G * N] ^= 1
        ref = O.logp_dlogp(theta, i_raw, w, coh, (20,))[0]
        assert abs(ctx.logp(0, theta) - ref) <= RTOL * abs(ref)


def test_argument_errors():
    from abdpymc_amd._native import Context

    coh = oracle_cohort_from_synth(synthetic.make_cohort(8, 10, seed=1))
    with pytest.raises(ValueError, match="ascending"):
        _ctx(coh, (5, 2))
    with pytest.raises(ValueError, match="largest split"):
        _ctx(coh, (11,))
    with pytest.raises(ValueError, match="not unique"):
        _ctx(coh, (3, 3))
    with pytest.raises(ValueError, match="positive"):
        _ctx(coh, (-1,))
    ctx = _ctx(coh)
    with pytest.raises(Exception, match="no discrete state"):
        ctx.logp(0, synthetic.theta_init(10))
    with pytest.raises(ValueError):
        ctx.set_discrete(0, np.full((10, 8), 2, dtype=np.int8), np.zeros(8, dtype=np.int8))
    with pytest.raises(ValueError):
        ctx.set_discrete(1, np.zeros((10, 8), dtype=np.int8), np.zeros(8, dtype=np.int8))


def test_out_of_support_is_not_an_error():
    coh = oracle_cohort_from_synth(synthetic.make_cohort(8, 10, seed=1))
    ctx = _ctx(coh)
    theta, i_raw, w = _state(coh, 19)
    ctx.set_discrete(0, i_raw, w)
    theta[13] = 800.0  # sigma = exp(800) = inf
    lp, g = ctx.logp_dlogp(0, theta)
    assert not np.isfinite(lp)


def test_stream_ordered_results_equal_synchronous():
    """Stream order: launches rotate over two streams with half-size grids and the fixed-order sum of a launch runs
    inside the next launch on its stream; results go through the device ring.  A synchronous call uses the full
    grid, the standalone sum and mapped memory.  Each form is bit-reproducible; across forms the launch shapes
    differ, so they agree to rounding."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(700, 90, seed=31))
    ctx = _ctx(coh, (40,), n_chains=4)
    for c in range(4):
        _, i_raw, w = _state(coh, 200 + c)
        ctx.set_discrete(c, i_raw, w)
    rng = np.random.default_rng(1)
    thetas = synthetic.theta_init(90) + 0.3 * rng.standard_normal((7, 4, 17))
    ids = np.arange(4)
    for k in range(7):
        ctx.enqueue(k, ids, thetas[k])
    ctx.wait()
    lp_a, g_a = ctx.fetch_many(np.arange(7), 4)
    for k in range(7):
        ctx.enqueue(10 + k, ids, thetas[k])
    ctx.wait()
    lp_r, g_r = ctx.fetch_many(10 + np.arange(7), 4)
    np.testing.assert_array_equal(lp_r, lp_a)  # same form again: same bits, whichever stream a launch lands on
    np.testing.assert_array_equal(g_r, g_a)
    for k in range(7):
        lp_s, g_s = ctx.logp_dlogp_batch(ids, thetas[k])
        np.testing.assert_allclose(lp_a[k], lp_s, rtol=1e-13)
        scale = np.abs(g_s).max(axis=1, keepdims=True)
        assert (np.abs(g_a[k] - g_s) / scale).max() < 1e-13
    # interleaving synchronous and stream-ordered calls keeps every result where it belongs
    ctx.enqueue(0, ids, thetas[3])
    lp_mid, _ = ctx.logp_dlogp_batch(ids[:2], thetas[5][:2])
    ctx.enqueue(1, ids, thetas[4])
    ctx.wait()
    lp_b, _ = ctx.fetch_many([0, 1], 4)
    np.testing.assert_array_equal(lp_b[0], lp_a[3])
    np.testing.assert_array_equal(lp_b[1], lp_a[4])
    assert abs(lp_mid[1] - lp_a[5][1]) <= 1e-12 * abs(lp_mid[1])  # 2-chain launch: other shape, equal to rounding


def test_synchronous_batch_never_returns_a_stale_row():
    """Each chain's 16 sums are written by their own workgroup in no particular order; the host must wait for
    every row's completion tag, not only the last row's (regression: row 0 could be read one call late)."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(64, 12, seed=3))
    ctx = _ctx(coh, (), n_chains=16)
    for c in range(16):
        _, i_raw, w = _state(coh, 300 + c)
        ctx.set_discrete(c, i_raw, w)
    rng = np.random.default_rng(4)
    thetas = synthetic.theta_init(12) + 0.3 * rng.standard_normal((2, 16, 17))
    ids = np.arange(16)
    want = [ctx.logp_dlogp_batch(ids, thetas[k]) for k in range(2)]
    want = [ctx.logp_dlogp_batch(ids, thetas[k]) for k in range(2)]  # second pass: every row has been overwritten
    for it in range(6000):
        k = it & 1
        lp, g = ctx.logp_dlogp_batch(ids, thetas[k])
        assert np.array_equal(lp, want[k][0]) and np.array_equal(g, want[k][1]), it


def test_stream_queue_probe_and_pipes():
    """abd_stream_queues: every stream gets a queue number, numbers are 0 .. n_queues - 1 in order of first appearance,
    a dense context rotates its stream-ordered launches over at most one stream per queue (and at most four), and the
    stream-ordered results do not depend on that (they equal the synchronous ones to rounding)."""
    from abdpymc_amd import synthetic
    from abdpymc_amd._native import Context

    sc = synthetic.make_cohort(640, 40, seed=2)
    ctx = Context(40, 640, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=4)
    q = ctx.stream_queues()
    assert len(q) == 8 and q[0] == 0 and min(q) == 0
    # the probe exchanges streams of crowded queues for spare ones on the others: with 4 queues the 8 streams sit 2 + 2 + 2 + 2
    # (3 + 2 + 2 + 1 as created); allow for a spare stream that was not there to be had
    counts = [q.count(v) for v in set(q)]
    assert max(counts) - min(counts) <= 1 or max(counts) <= 3, q
    seen = []
    for v in q:
        if v not in seen:
            assert v == len(seen)
            seen.append(v)
    assert 1 <= ctx.n_pipes <= min(4, len(seen))
    for c in range(4):
        ctx.set_discrete(c, *synthetic.make_chain_state(640, 40, c))
    th = np.stack([synthetic.make_thetas(40, 1, c)[0] for c in range(4)])
    for k in range(12):
        ctx.enqueue(k, np.arange(4), th + 0.01 * k)
    ctx.wait()
    lp, g = ctx.fetch_many(np.arange(12), 4)
    for k in range(12):
        lp_s, g_s = ctx.logp_dlogp_batch(np.arange(4), th + 0.01 * k)
        np.testing.assert_allclose(lp[k], lp_s, rtol=1e-12)
        np.testing.assert_allclose(g[k], g_s, rtol=1e-9, atol=1e-9 * np.abs(g_s).max())
    ctx.close()


def test_many_evaluations_in_one_call():
    """abd_logp_dlogp_many = enqueue every step, wait once, fetch: the same bits as doing that by hand, for more steps
    than the result ring has slots too, and logp-only when no gradient is asked for."""
    from abdpymc_amd import synthetic
    from abdpymc_amd._native import Context

    sc = synthetic.make_cohort(400, 30, seed=3)
    ctx = Context(30, 400, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, n_chains=2)
    for c in range(2):
        ctx.set_discrete(c, *synthetic.make_chain_state(400, 30, c))
    K = ctx.n_result_slots + 37
    th = np.stack([np.stack([synthetic.make_thetas(30, 1, 7 * k + c)[0] for c in range(2)]) for k in range(K)])
    lp, g = ctx.logp_dlogp_many([0, 1], th)
    assert lp.shape == (K, 2) and g.shape == (K, 2, 17)
    for k in (0, 1, ctx.n_result_slots - 1, ctx.n_result_slots, K - 1):
        ctx.enqueue(0, [0, 1], th[k])
        ctx.wait()
        lp1, g1 = ctx.fetch(0, 2)
        np.testing.assert_array_equal(lp[k], lp1)
        np.testing.assert_array_equal(g[k], g1)
    with pytest.raises(ValueError):
        ctx.logp_dlogp_many([0, 1], th[:, :1])
    ctx.close()


@pytest.mark.parametrize("lanes", ["1", "0"])
@pytest.mark.parametrize("G,splits", [(300, (100, 201)), (512, (256,)), (449, None)])
def test_more_than_256_gaps_observation_lists_and_deterministics(G, splits, lanes, monkeypatch):
    """The reference takes any n_gaps (abd.py:101, 224-239).  Beyond 256 gaps the kernels that hold an individual's gap
    axis in registers run in their 8-word form: both observation-list kernels, the Deterministics, the flip path."""
    monkeypatch.setenv("ABD_OBS_LANES", lanes)
    coh = random_sparse_cohort(29, G, 1500, 1200, seed=G)
    ctx = _ctx(coh, splits)
    assert not ctx.is_dense
    theta, i_raw, w = _state(coh, 3)
    i_raw[-1, :5] = 1  # the last gap of the axis
    ctx.set_discrete(0, i_raw, w)
    lp, g = ctx.logp_dlogp(0, theta)
    lp_ref, g_ref = O.logp_dlogp(theta, i_raw, w, coh, splits)
    assert_close(lp, g, lp_ref, g_ref)
    i_dev, mun, mus = ctx.deterministics(0, theta)
    i_ref = O.constrain_infections(i_raw, np.asarray(coh.pcrpos).T, splits)
    np.testing.assert_array_equal(i_dev, i_ref)
    # one flipped bit in the upper words: the slot's cached constrained words and counters follow
    flat = (G - 2) * coh.n_inds + 7
    ctx.flip_discrete(0, flat)
    i_raw.ravel()[flat] ^= 1
    lp2, g2 = ctx.logp_dlogp(0, theta)
    assert_close(lp2, g2, *O.logp_dlogp(theta, i_raw, w, coh, splits))
    ctx.close()


def test_one_chain_launches_read_the_split_panels_and_agree_with_the_pair_panels():
    """A launch that evaluates ONE chain reads od and a one-byte code of the cell's log dilution (<= 256 distinct values per
    antigen) instead of {od, log_dilution} pairs: the same numbers to rounding as the batched launch over the pair panels,
    both within 1e-6 of the oracle; with more distinct dilutions than a byte can code the library stays on the pair panels."""
    sc = synthetic.make_cohort(333, 97, seed=12)
    coh = oracle_cohort_from_synth(sc)
    for many_dilutions in (False, True):
        if many_dilutions:  # every cell its own dilution: no dictionary of 256 entries holds them
            rng = np.random.default_rng(1)
            coh.s.log_dilution[:] = rng.uniform(0.0, 4.0, coh.s.log_dilution.size)
        ctx = _ctx(coh, (30, 61), n_chains=4)
        assert ctx.is_dense
        cells = 97 * 333
        # algorithmic bytes of a one-chain launch: (8 + 1) vs 16 bytes per cell and antigen
        nb = ctx.algorithmic_bytes(1)
        assert (cells * 32 <= nb < cells * 32 + 40000) if many_dilutions else (cells * 18 <= nb < cells * 18 + 40000)
        st = [_state(coh, 20 + c) for c in range(4)]
        for c, (_, i_raw, w) in enumerate(st):
            ctx.set_discrete(c, i_raw, w)
        thetas = np.array([t for t, _, _ in st])
        lp4, g4 = ctx.logp_dlogp_batch([0, 1, 2, 3], thetas)  # one launch, pair panels
        for c in range(4):
            lp1, g1 = ctx.logp_dlogp(c, thetas[c])  # one chain per launch
            ref = O.logp_dlogp(thetas[c], st[c][1], st[c][2], coh, (30, 61))
            assert_close(lp1, g1, *ref)
            assert_close(lp4[c], g4[c], *ref)
            assert abs(lp1 - lp4[c]) <= 1e-12 * abs(lp1)
        ctx.close()


def test_fp32_storage_split_panels_hold_what_the_pair_panels_hold():
    """fp32 storage with log dilutions that fp32 does not hold exactly (log10 of the assay's dilutions): the one-chain launch
    (split panels: od + dictionary code) and the batched launch (pair panels) both see the fp32-rounded values, so they
    agree to rounding with each other and with the oracle on the rounded panels."""
    sc = synthetic.make_cohort(150, 70, seed=31)
    coh = oracle_cohort_from_synth(sc)
    rng = np.random.default_rng(2)
    dil = np.log10(np.array([100.0, 300.0, 640.0, 2000.0, 5000.0]))
    coh.s.log_dilution[:] = rng.choice(dil, coh.s.log_dilution.size)
    coh.n.log_dilution[:] = rng.choice(dil, coh.n.log_dilution.size)
    ctx = _ctx(coh, (20, 45), n_chains=2, storage="f32")
    assert ctx.is_dense
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    coh32 = O.Cohort(coh.n_gaps, coh.n_inds, coh.vacs, coh.pcrpos,
                     O.AntigenObs(coh.s.idx_gap, coh.s.idx_ind, r32(coh.s.log_dilution), r32(coh.s.od)),
                     O.AntigenObs(coh.n.idx_gap, coh.n.idx_ind, r32(coh.n.log_dilution), r32(coh.n.od)))
    th = []
    for c in range(2):
        theta, i_raw, w = _state(coh, 40 + c)
        ctx.set_discrete(c, i_raw, w)
        th.append((theta, i_raw, w))
    lp_b, g_b = ctx.logp_dlogp_batch(np.arange(2, dtype=np.int32), np.stack([t[0] for t in th]))  # pair panels
    for c in range(2):
        lp1, g1 = ctx.logp_dlogp(c, th[c][0])  # split panels
        lp_ref, g_ref = O.logp_dlogp(th[c][0], th[c][1], th[c][2], coh32, (20, 45))
        assert_close(lp1, g1, lp_ref, g_ref)
        assert_close(lp_b[c], g_b[c], lp_ref, g_ref)
        assert abs(lp1 - lp_b[c]) <= 1e-12 * abs(lp_ref)
    ctx.close()
