"""
The binary Gibbs-Metropolis sweep.

CPU: Philox4x32-10 known answers; the per-individual sweep of oracle/abd_oracle.c equals a literal
BinaryGibbsMetropolis restatement that evaluates the FULL joint logp for every proposal (the reference's
semantics, abd.py:922) when both consume the same random stream -- i.e. the factorisation the device kernel
relies on is exact.
GPU: abd_gibbs_sweep reproduces the CPU restatement's trajectory bit for bit.
"""
import math

import numpy as np
import pytest

from abdpymc_amd import synthetic
from oracle import abd_oracle as O
from oracle import c_oracle
from tests.helpers import oracle_cohort_from_synth, random_sparse_cohort

TRANSIT_U32 = 3435973836


def _state(coh, seed, rate=None):
    rng = np.random.default_rng(seed)
    rate = rate if rate is not None else 2.0 / coh.n_gaps
    i_raw = (rng.random((coh.n_gaps, coh.n_inds)) < rate).astype(np.int8)
    w = (rng.random(coh.n_inds) < 0.5).astype(np.int8)
    theta = synthetic.theta_init(coh.n_gaps) + 0.3 * rng.standard_normal(17)
    return theta, i_raw, w


def test_philox_known_answers():
    co = c_oracle.COracle(oracle_cohort_from_synth(synthetic.make_cohort(3, 4, seed=1)))
    # Random123 kat_vectors, philox4x32-10
    assert co.philox(0, 0, 0, 0, 0, 0) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert co.philox(*[0xFFFFFFFF] * 6) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert co.philox(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def literal_sweep(co, coh, theta, i_raw, waner, chain, seed, sweep, splits=None, ignore=False):
    """BinaryGibbsMetropolis.astep with the JOINT logp evaluated for every proposal, same random stream."""
    i_raw, waner = i_raw.copy(), waner.copy()
    G, N = coh.n_gaps, coh.n_inds
    k0 = (seed & 0xFFFFFFFF) ^ ((sweep * 0x9E3779B9) & 0xFFFFFFFF)
    k1 = (seed >> 32) & 0xFFFFFFFF
    lp = O.joint_logp(theta, i_raw, waner, coh, splits, ignore, dense=False)
    acc = prop = 0
    for j in range(N):  # individuals commute; any cross-individual order gives the same result
        r = [co.philox(d, j, chain, 0, k0, k1) for d in range(G + 1)]
        keys = [(r[d][0] & ~0x1FF) | d for d in range(G + 1)]
        for d in sorted(range(G + 1), key=lambda d: keys[d]):
            if not r[d][1] < TRANSIT_U32:
                continue
            prop += 1
            if d < G:
                i_raw[d, j] ^= 1
            else:
                waner[j] ^= 1
            lp_new = O.joint_logp(theta, i_raw, waner, coh, splits, ignore, dense=False)
            u = (r[d][2] + 0.5) / 4294967296.0
            delta = lp_new - lp
            if delta > 0 or delta > math.log(u):
                lp = lp_new
                acc += 1
            elif d < G:
                i_raw[d, j] ^= 1
            else:
                waner[j] ^= 1
    return i_raw, waner, acc, prop


@pytest.mark.parametrize("splits", [None, (5,), (4, 9)])
def test_per_individual_sweep_equals_joint_logp_sweep(splits):
    coh = oracle_cohort_from_synth(synthetic.make_cohort(7, 13, seed=4))
    co = c_oracle.COracle(coh, splits)
    theta, i_raw, w = _state(coh, 3, rate=0.2)
    a = co.gibbs_sweep(theta, i_raw, w, chain=1, seed=0x1234567812345678, sweep=5, nthreads=2)
    b = literal_sweep(co, coh, theta, i_raw, w, 1, 0x1234567812345678, 5, splits)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert (a[2], a[3]) == (b[2], b[3])
    assert 0 < a[2] <= a[3] <= 7 * 14
    assert abs(a[3] / (7 * 14) - 0.8) < 0.15


def test_sweep_sparse_cohort_literal():
    coh = random_sparse_cohort(6, 11, 60, 50, seed=8)
    co = c_oracle.COracle(coh, (6,))
    theta, i_raw, w = _state(coh, 9, rate=0.3)
    a = co.gibbs_sweep(theta, i_raw, w, chain=0, seed=77, sweep=0)
    b = literal_sweep(co, coh, theta, i_raw, w, 0, 77, 0, (6,))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert (a[2], a[3]) == (b[2], b[3])


def test_sweep_targets_the_conditional():
    """Long run on one tiny individual: empirical frequencies of waner match its exact conditional."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(1, 6, seed=12))
    co = c_oracle.COracle(coh)
    theta, i_raw, w = _state(coh, 1, rate=0.0)
    # enumerate the exact joint over (i_raw column, waner): 2^7 states
    states, lps = [], []
    for m in range(128):
        ir = np.array([(m >> g) & 1 for g in range(6)], dtype=np.int8)[:, None]
        ww = np.array([(m >> 6) & 1], dtype=np.int8)
        states.append((ir, ww))
        lps.append(O.joint_logp(theta, ir, ww, coh, dense=False))
    lps = np.array(lps)
    pr = np.exp(lps - lps.max())
    pr /= pr.sum()
    p_w1 = pr[[m for m in range(128) if (m >> 6) & 1]].sum()
    cnt = 0
    n = 4000
    for s in range(n):
        i_raw, w, _, _ = co.gibbs_sweep(theta, i_raw, w, chain=0, seed=5, sweep=s)
        cnt += int(w[0])
    assert abs(cnt / n - p_w1) < 4 * math.sqrt(p_w1 * (1 - p_w1) / n) + 0.02


# ---------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _ctx(coh, splits=None, ignore=False, n_chains=1, storage="f64"):
    from abdpymc_amd._native import Context

    return Context(coh.n_gaps, coh.n_inds, (coh.s.idx_gap, coh.s.idx_ind, coh.s.log_dilution, coh.s.od),
                   (coh.n.idx_gap, coh.n.idx_ind, coh.n.log_dilution, coh.n.od), coh.vacs,
                   None if ignore else coh.pcrpos, splits=splits, n_chains=n_chains, storage=storage)


@gpu
@pytest.mark.parametrize("G,N,splits", [(20, 23, None), (70, 130, (30,)), (200, 64, (66, 133)), (256, 9, None), (5, 300, (2,)),
                                        (300, 40, (100, 200)), (512, 11, None), (257, 30, (0, 257))])  # beyond 256 gaps: 8 words per individual
def test_gpu_sweep_matches_cpu_restatement_dense(G, N, splits):
    coh = oracle_cohort_from_synth(synthetic.make_cohort(N, G, seed=G + N))
    co = c_oracle.COracle(coh, splits)
    ctx = _ctx(coh, splits, n_chains=2)
    thetas, states = [], []
    for c in range(2):
        theta, i_raw, w = _state(coh, 40 + c)
        ctx.set_discrete(c, i_raw, w)
        thetas.append(theta)
        states.append((i_raw, w))
    for sweep in range(3):
        acc, prop = ctx.gibbs_sweep([0, 1], np.array(thetas), seed=2024, sweep=sweep)
        for c in range(2):
            i_ref, w_ref, a_ref, p_ref = co.gibbs_sweep(thetas[c], states[c][0], states[c][1], chain=c, seed=2024, sweep=sweep)
            i_gpu, w_gpu = ctx.get_discrete(c)
            np.testing.assert_array_equal(i_gpu, i_ref)
            np.testing.assert_array_equal(w_gpu, w_ref)
            assert (int(acc[c]), int(prop[c])) == (a_ref, p_ref)
            states[c] = (i_ref, w_ref)
    # the evaluation path sees the updated state
    lp, _ = ctx.logp_dlogp(1, thetas[1])
    ref = O.logp_dlogp(thetas[1], states[1][0], states[1][1], coh, splits)[0]
    assert abs(lp - ref) <= 1e-6 * abs(ref)


@gpu
@pytest.mark.parametrize("G,N,splits,rate", [(200, 40, None, 1.0), (300, 24, (100, 200), 1.0), (512, 9, (256,), 0.5), (64, 50, None, 0.9)])
def test_gpu_sweep_dense_states_full_of_infections(G, N, splits, rate):
    """Raw states with (nearly) every gap set: the kept infections are one in four gaps, far more than a lane of the
    lane-per-proposal kernel holds for its walk (ABD_G2_KCAP) -- those proposals are evaluated by the whole wave at the
    frontier.  Same trajectories as the CPU restatement, and as the wave-per-proposal kernel."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(N, G, seed=3 * G + N))
    co = c_oracle.COracle(coh, splits)
    ctx = _ctx(coh, splits, n_chains=1)
    theta, i_raw, w = _state(coh, 77, rate=rate)
    ctx.set_discrete(0, i_raw, w)
    for sweep in range(2):
        acc, prop = ctx.gibbs_sweep([0], theta[None, :], seed=5, sweep=sweep)
        i_ref, w_ref, a_ref, p_ref = co.gibbs_sweep(theta, i_raw, w, chain=0, seed=5, sweep=sweep)
        i_gpu, w_gpu = ctx.get_discrete(0)
        np.testing.assert_array_equal(i_gpu, i_ref)
        np.testing.assert_array_equal(w_gpu, w_ref)
        assert (int(acc[0]), int(prop[0])) == (a_ref, p_ref)
        i_raw, w = i_ref, w_ref
    ctx.close()


@gpu
@pytest.mark.parametrize("ignore", [False, True])
@pytest.mark.parametrize("N,G,ks,kn,splits", [(41, 26, 900, 700, (10,)), (12, 200, 2500, 2100, (66, 133)), (30, 70, 40, 0, None),
                                              (9, 300, 900, 800, (150,)), (7, 512, 600, 700, None)])
def test_gpu_sweep_matches_cpu_restatement_sparse(ignore, N, G, ks, kn, splits):
    coh = random_sparse_cohort(N, G, ks, kn, seed=5)
    co = c_oracle.COracle(coh, splits, ignore)
    ctx = _ctx(coh, splits, ignore)
    theta, i_raw, w = _state(coh, 6)
    ctx.set_discrete(0, i_raw, w)
    for sweep in range(2):
        acc, prop = ctx.gibbs_sweep([0], theta[None], seed=99, sweep=sweep)
        i_raw, w, a_ref, p_ref = co.gibbs_sweep(theta, i_raw, w, chain=0, seed=99, sweep=sweep)
        i_gpu, w_gpu = ctx.get_discrete(0)
        np.testing.assert_array_equal(i_gpu, i_raw)
        np.testing.assert_array_equal(w_gpu, w)
        assert (int(acc[0]), int(prop[0])) == (a_ref, p_ref)


@gpu
def test_gpu_sweep_full_size_10000x200():
    import time

    sc = synthetic.make_cohort(10000, 200)
    coh = oracle_cohort_from_synth(sc)
    ctx = _ctx(coh, n_chains=4)
    thetas = []
    for c in range(4):
        ctx.set_discrete(c, *synthetic.make_chain_state(10000, 200, c))
        thetas.append(synthetic.make_thetas(200, 1, c)[0])
    t0 = time.perf_counter()
    acc, prop = ctx.gibbs_sweep(np.arange(4), np.array(thetas), seed=1, sweep=0)
    dt = time.perf_counter() - t0
    print(f"\nfull-size Gibbs sweep, 4 chains x 2.01 M dims: {dt * 1e3:.1f} ms, accepted {acc.tolist()} of {prop.tolist()}")
    assert np.all(np.abs(prop / (10000 * 201) - 0.8) < 0.01)
    # chain 2 against the CPU restatement (OpenMP)
    co = c_oracle.COracle(coh)
    i0, w0 = synthetic.make_chain_state(10000, 200, 2)
    i_ref, w_ref, a_ref, p_ref = co.gibbs_sweep(thetas[2], i0, w0, chain=2, seed=1, sweep=0, nthreads=8)
    i_gpu, w_gpu = ctx.get_discrete(2)
    assert (int(acc[2]), int(prop[2])) == (a_ref, p_ref)
    np.testing.assert_array_equal(i_gpu, i_ref)
    np.testing.assert_array_equal(w_gpu, w_ref)


@gpu
def test_gpu_sweep_streams_are_keyed_by_chain_slot():
    """A chain's sweep depends on its slot id, not on its position in the call: sweeping slot 2 alone, listed
    second, or in the second launch group of a 20-chain call gives the same bits; equal states in different
    slots get different proposals."""
    coh = oracle_cohort_from_synth(synthetic.make_cohort(40, 30, seed=8))
    co = c_oracle.COracle(coh, None)
    theta, i_raw, w = _state(coh, 5)
    ctx = _ctx(coh, n_chains=20)
    refs = {c: co.gibbs_sweep(theta, i_raw, w, chain=c, seed=9, sweep=4) for c in (0, 2, 17)}
    assert not np.array_equal(refs[0][0], refs[2][0]) and not np.array_equal(refs[0][0], refs[17][0])
    for order in ([2], [5, 2], list(range(20))):
        for c in range(20):
            ctx.set_discrete(c, i_raw, w)
        ctx.gibbs_sweep(order, np.tile(theta, (len(order), 1)), seed=9, sweep=4)
        for c in (2, 17) if len(order) == 20 else (2,):
            i_gpu, w_gpu = ctx.get_discrete(c)
            np.testing.assert_array_equal(i_gpu, refs[c][0])
            np.testing.assert_array_equal(w_gpu, refs[c][1])


@gpu
@pytest.mark.parametrize("dense", [True, False])
def test_gpu_sweep_fp32_storage(dense):
    """fp32-held panels / lists: the sweep equals the CPU restatement run on the fp32-rounded data."""
    if dense:
        coh = oracle_cohort_from_synth(synthetic.make_cohort(70, 90, seed=12))
    else:
        from tests.helpers import random_sparse_cohort

        coh = random_sparse_cohort(40, 30, 900, 800, seed=13)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    coh32 = O.Cohort(coh.n_gaps, coh.n_inds, coh.vacs, coh.pcrpos,
                     O.AntigenObs(coh.s.idx_gap, coh.s.idx_ind, r32(coh.s.log_dilution), r32(coh.s.od)),
                     O.AntigenObs(coh.n.idx_gap, coh.n.idx_ind, r32(coh.n.log_dilution), r32(coh.n.od)))
    co = c_oracle.COracle(coh32, (20,))
    ctx = _ctx(coh, (20,), n_chains=1, storage="f32")
    assert ctx.is_dense == dense
    theta, i_raw, w = _state(coh, 77)
    ctx.set_discrete(0, i_raw, w)
    for sweep in range(2):
        acc, prop = ctx.gibbs_sweep([0], theta[None], seed=31, sweep=sweep)
        i_raw, w, a_ref, p_ref = co.gibbs_sweep(theta, i_raw, w, chain=0, seed=31, sweep=sweep)
        i_gpu, w_gpu = ctx.get_discrete(0)
        np.testing.assert_array_equal(i_gpu, i_raw)
        np.testing.assert_array_equal(w_gpu, w)
        assert (int(acc[0]), int(prop[0])) == (a_ref, p_ref)
