"""
BASELINE.json's configurations at full size, on the GPU, through the C ABI:
  config 2  synthetic 1 000 x 60,  fp64, 4 chains  -- against the NumPy oracle
  config 3  synthetic 10 000 x 200, fp64, 4 chains -- against the plain-C port (OpenMP) and through
            size-independent properties: the data term is additive over individuals (two half cohorts sum to
            the whole), a batched launch equals four single launches, results do not depend on the grid
  config 5  one GPU's share at full size: 100 000 x 200, fp32 storage, 1 chain (and a 20 000 x 200 case)
"""
import numpy as np
import pytest

from abdpymc_amd import synthetic
from oracle import abd_oracle as O
from oracle import c_oracle
from tests.helpers import oracle_cohort_from_synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _ctx(sc, n_chains, storage="f64", splits=None, sub=None):
    from abdpymc_amd._native import Context

    if sub is None:
        return Context(sc.n_gaps, sc.n_inds, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=splits, n_chains=n_chains, storage=storage)
    lo, hi = sub
    G = sc.n_gaps
    m = (sc.idx_ind >= lo) & (sc.idx_ind < hi)
    obs = lambda x, y: (sc.idx_gap[m], sc.idx_ind[m] - lo, x[m], y[m])  # noqa: E731
    return Context(G, hi - lo, obs(sc.x_s, sc.y_s), obs(sc.x_n, sc.y_n), sc.vacs[lo:hi], sc.pcrpos[lo:hi], splits=splits,
                   n_chains=n_chains, storage=storage)


def _close(lp, g, lp_ref, g_ref, rtol=RTOL):
    assert abs(lp - lp_ref) <= rtol * abs(lp_ref), (lp, lp_ref)
    scale = np.maximum(np.abs(g_ref), 1e-6 * np.abs(g_ref).max())
    assert (np.abs(np.asarray(g) - g_ref) / scale).max() <= rtol


def test_config2_1000x60_4chains():
    sc = synthetic.make_cohort(1000, 60)
    coh = oracle_cohort_from_synth(sc)
    ctx = _ctx(sc, 4)
    thetas = []
    for c in range(4):
        ctx.set_discrete(c, *synthetic.make_chain_state(1000, 60, c))
        thetas.append(synthetic.make_thetas(60, 1, c)[0])
    lp, g = ctx.logp_dlogp_batch(np.arange(4), np.array(thetas))
    for c in range(4):
        i_raw, w = synthetic.make_chain_state(1000, 60, c)
        _close(lp[c], g[c], *O.logp_dlogp(thetas[c], i_raw, w, coh))


@pytest.fixture(scope="module")
def c3():
    sc = synthetic.make_cohort(10000, 200)
    return sc, oracle_cohort_from_synth(sc)


@pytest.mark.parametrize("splits", [None, (100,), (66, 133)])
def test_config3_10000x200_vs_c_port(c3, splits):
    sc, coh = c3
    ctx = _ctx(sc, 4, splits=splits)
    co = c_oracle.COracle(coh, splits)
    thetas = []
    for c in range(4):
        ctx.set_discrete(c, *synthetic.make_chain_state(10000, 200, c))
        thetas.append(synthetic.make_thetas(200, 1, c)[0])
    lp, g = ctx.logp_dlogp_batch(np.arange(4), np.array(thetas))
    for c in (0, 3):
        i_raw, w = synthetic.make_chain_state(10000, 200, c)
        _close(lp[c], g[c], *co.logp_dlogp(thetas[c], i_raw, w, nthreads=8))
    # batched launch == single launches (different launch shape: equal to rounding)
    for c in range(4):
        lp1, g1 = ctx.logp_dlogp(c, thetas[c])
        assert abs(lp1 - lp[c]) <= 1e-12 * abs(lp[c])
        np.testing.assert_allclose(g1, g[c], rtol=1e-9, atol=1e-9 * np.abs(g[c]).max())
    # grid independence
    ctx.set_launch_config(333, 0)
    lp2, g2 = ctx.logp_dlogp_batch(np.arange(4), np.array(thetas))
    np.testing.assert_allclose(lp2, lp, rtol=1e-12)
    np.testing.assert_allclose(g2, g, rtol=1e-9, atol=1e-9 * np.abs(g).max())
    ctx.close()


def test_config3_data_term_is_additive_over_individuals(c3):
    sc, _ = c3
    i_raw, w = synthetic.make_chain_state(10000, 200, 0)
    theta = synthetic.make_thetas(200, 1, 0)[0]
    full = _ctx(sc, 1)
    full.set_discrete(0, i_raw, w)
    ll, g = full.loglik_dlogp(0, theta)
    parts = []
    for lo, hi in ((0, 4100), (4100, 10000)):
        part = _ctx(sc, 1, sub=(lo, hi))
        part.set_discrete(0, i_raw[:, lo:hi], w[lo:hi])
        parts.append(part.loglik_dlogp(0, theta))
        part.close()
    assert abs((parts[0][0] + parts[1][0]) - ll) <= 1e-11 * abs(ll)
    np.testing.assert_allclose(parts[0][1] + parts[1][1], g, rtol=1e-8, atol=1e-9 * np.abs(g).max())
    full.close()


def test_config5_fp32_storage_20000x200():
    sc = synthetic.make_cohort(20000, 200, seed=5)
    coh = oracle_cohort_from_synth(sc)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
    coh32 = O.Cohort(coh.n_gaps, coh.n_inds, coh.vacs, coh.pcrpos,
                     O.AntigenObs(coh.s.idx_gap, coh.s.idx_ind, r32(coh.s.log_dilution), r32(coh.s.od)),
                     O.AntigenObs(coh.n.idx_gap, coh.n.idx_ind, r32(coh.n.log_dilution), r32(coh.n.od)))
    ctx = _ctx(sc, 1, storage="f32")
    i_raw, w = synthetic.make_chain_state(20000, 200, 0)
    theta = synthetic.make_thetas(200, 1, 0)[0]
    ctx.set_discrete(0, i_raw, w)
    lp, g = ctx.logp_dlogp(0, theta)
    co = c_oracle.COracle(coh32)
    _close(lp, g, *co.logp_dlogp(theta, i_raw, w, nthreads=8))  # fp64 arithmetic on the fp32-held panels
    co64 = c_oracle.COracle(coh)
    lp64, _ = co64.logp_dlogp(theta, i_raw, w, nthreads=8)
    assert abs(lp - lp64) <= 1e-4 * abs(lp64)  # storage rounding only: tolerance 1e-4 relative, stated


def test_config5_full_size_100000x200_fp32():
    """BASELINE config 5 at its real size -- one GPU's share: 100 000 x 200, fp32 storage, 1 chain (330 MB of panels,
    beyond the 256 MiB Infinity Cache): against the plain-C port on the fp32-rounded panels, and additive over two
    unequal parts of the cohort."""
    N, G = 100000, 200
    sc = synthetic.make_cohort(N, G)
    coh = oracle_cohort_from_synth(sc)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
    coh32 = O.Cohort(coh.n_gaps, coh.n_inds, coh.vacs, coh.pcrpos,
                     O.AntigenObs(coh.s.idx_gap, coh.s.idx_ind, r32(coh.s.log_dilution), r32(coh.s.od)),
                     O.AntigenObs(coh.n.idx_gap, coh.n.idx_ind, r32(coh.n.log_dilution), r32(coh.n.od)))
    i_raw, w = synthetic.make_chain_state(N, G, 0)
    thetas = synthetic.make_thetas(G, 2, 0)
    ctx = _ctx(sc, 1, storage="f32")
    assert ctx.is_dense
    ctx.set_discrete(0, i_raw, w)
    co = c_oracle.COracle(coh32)
    for theta in thetas:
        lp, g = ctx.logp_dlogp(0, theta)
        _close(lp, g, *co.logp_dlogp(theta, i_raw, w, nthreads=16))
    # stream-ordered launches (1 workgroup per CU, three in flight) give the same numbers as the synchronous call
    lp_sync, g_sync = ctx.logp_dlogp(0, thetas[0])
    for k in range(6):
        ctx.enqueue(k, [0], thetas[k % 2][None])
    ctx.wait()
    lp_q, g_q = ctx.fetch_many(np.arange(6), 1)
    np.testing.assert_allclose(lp_q[0, 0], lp_sync, rtol=1e-12)
    np.testing.assert_allclose(lp_q[4, 0], lp_sync, rtol=1e-12)
    np.testing.assert_allclose(g_q[2, 0], g_sync, rtol=1e-9, atol=1e-9 * np.abs(g_sync).max())
    # the data term is a sum over individuals: two parts of the cohort add up to the whole
    ll, gl = ctx.loglik_dlogp(0, thetas[0])
    ctx.close()
    parts = []
    for lo, hi in ((0, 37000), (37000, N)):
        part = _ctx(sc, 1, storage="f32", sub=(lo, hi))
        part.set_discrete(0, i_raw[:, lo:hi], w[lo:hi])
        parts.append(part.loglik_dlogp(0, thetas[0]))
        part.close()
    assert abs((parts[0][0] + parts[1][0]) - ll) <= 1e-11 * abs(ll)
    np.testing.assert_allclose(parts[0][1] + parts[1][1], gl, rtol=1e-8, atol=1e-9 * np.abs(gl).max())
