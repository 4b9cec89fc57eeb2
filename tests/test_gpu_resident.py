"""
The resident evaluation kernel of the native sampler (abd_resident.hpp, opt-in with ABD_RESIDENT=1): on a dense cohort
with one chain per unit the evaluation kernel is launched once per NUTS trajectory and fed the leapfrogs' thetas through
mapped host memory.

Pinned here:
 * a sampler run with resident kernels gives the SAME BITS as the same run with one launch per evaluation
   (ABD_RESIDENT=0): same range table, same walk, same fixed-order sum;
 * a kernel that times out for lack of commands is relaunched and the chain does not notice (forced with a 1 us
   time-out: every wait between two leapfrogs expires);
 * the kernels are really used (abd_resident_stats), nothing is relaunched on an idle machine, and the run ends with
   no kernel left behind (the context is destroyed and re-created around every run).
"""
import numpy as np
import pytest

from abdpymc_amd import synthetic

pytestmark = pytest.mark.gpu


def _run(monkeypatch, N, G, C, iters, env, splits=(), gibbs=True, storage="f64", tune=10):
    from abdpymc_amd._native import Context

    for k in ("ABD_RESIDENT", "ABD_RESIDENT_TIMEOUT_MS", "ABD_SAMPLER_UNIT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = synthetic.make_cohort(N, G, seed=4)
    ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=splits, n_chains=C, storage=storage)
    q0 = np.stack([synthetic.make_thetas(G, 1, c)[0] for c in range(C)])
    for c in range(C):
        ctx.set_discrete(c, *synthetic.make_chain_state(N, G, c))
    smp = ctx.sampler(np.arange(C), q0, tune=tune, seed=5, gibbs=gibbs)
    th, st = smp.run(iters)
    smp.close()
    states = [ctx.get_discrete(c) for c in range(C)]
    stats = ctx.resident_stats
    fb = ctx.wait_fallbacks
    ctx.close()
    return th, st, states, stats, fb


@pytest.mark.parametrize("N,G,splits,storage", [(300, 40, (17,), "f64"), (1000, 60, (), "f64"), (700, 130, (40, 90), "f32")])
def test_resident_kernels_give_the_bits_of_launched_evaluations(monkeypatch, N, G, splits, storage):
    C, iters = 3, 12
    base = {"ABD_SAMPLER_UNIT": "1"}
    th0, st0, s0, stats0, _ = _run(monkeypatch, N, G, C, iters, dict(base, ABD_RESIDENT="0"), splits, storage=storage)
    th1, st1, s1, stats1, fb1 = _run(monkeypatch, N, G, C, iters, dict(base, ABD_RESIDENT="1"), splits, storage=storage)
    assert stats0 == {"launches": 0, "commands": 0, "restarts": 0}
    leapfrogs = int(st1["n_steps"].sum())
    assert stats1["launches"] == C * iters and stats1["commands"] == leapfrogs and stats1["restarts"] == 0, (stats1, leapfrogs)
    assert fb1 == 0
    np.testing.assert_array_equal(th0, th1)
    for k in st0:
        np.testing.assert_array_equal(st0[k], st1[k], err_msg=k)
    for (a, b), (c, d) in zip(s0, s1):
        np.testing.assert_array_equal(a, c)
        np.testing.assert_array_equal(b, d)


def test_a_kernel_that_times_out_is_relaunched_unnoticed(monkeypatch):
    N, G, C, iters = 300, 40, 2, 6
    base = {"ABD_SAMPLER_UNIT": "1"}
    th0, st0, s0, _, _ = _run(monkeypatch, N, G, C, iters, dict(base, ABD_RESIDENT="0"), (17,))
    th1, st1, s1, stats1, _ = _run(monkeypatch, N, G, C, iters, dict(base, ABD_RESIDENT="1", ABD_RESIDENT_TIMEOUT_MS="0.001"), (17,))
    assert stats1["restarts"] > 0, stats1
    np.testing.assert_array_equal(th0, th1)
    np.testing.assert_array_equal(st0["lp"], st1["lp"])
    for (a, b), (c, d) in zip(s0, s1):
        np.testing.assert_array_equal(a, c)
        np.testing.assert_array_equal(b, d)


def test_continuous_only_trajectories_at_full_size(monkeypatch):
    """Config 3's shape (10 000 x 200), 4 chains without the sweep: resident and launched runs agree bit for bit; the
    resident kernels are opt-in (the default run launches every evaluation)."""
    N, G, C, iters = 10000, 200, 4, 6
    th0, st0, _, stats0, _ = _run(monkeypatch, N, G, C, iters, {}, gibbs=False, tune=10 ** 6)
    th1, st1, _, stats1, fb = _run(monkeypatch, N, G, C, iters, {"ABD_RESIDENT": "1"}, gibbs=False, tune=10 ** 6)
    assert stats0["launches"] == 0 and stats1["launches"] == C * iters and stats1["restarts"] == 0 and fb == 0, (stats0, stats1, fb)
    np.testing.assert_array_equal(th0, th1)
    np.testing.assert_array_equal(st0["lp"], st1["lp"])
    np.testing.assert_array_equal(st0["n_steps"], st1["n_steps"])
