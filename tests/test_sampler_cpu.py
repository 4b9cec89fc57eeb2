"""The sampler machinery on closed-form targets (no GPU): NUTS on a correlated-scale Gaussian, binary Gibbs on
independent Bernoullis, and the CLI's flag set (reference abd.py:888-911)."""
import math

import numpy as np
import pytest

from abdpymc_amd.cli import build_parser
from abdpymc_amd.sampler import DualAveraging, Nuts, binary_gibbs_sweep


def test_nuts_recovers_gaussian_moments():
    rng = np.random.default_rng(1)
    mu = np.array([1.0, -2.0, 0.5])
    sd = np.array([0.5, 2.0, 1.0])

    def fn(q):
        z = (q - mu) / sd
        return -0.5 * float(z @ z), -z / sd

    nuts = Nuts(fn, 3, rng)
    q = np.zeros(3)
    lp, g = fn(q)
    nuts.eps = nuts.find_reasonable_eps(q, lp, g)
    nuts.da = DualAveraging(nuts.eps)
    xs = []
    for it in range(1500):
        q, lp, g, st = nuts.step(q, lp, g, adapt=it < 500)
        if it == 499:
            nuts.eps = nuts.da.final()
        if it >= 500:
            xs.append(q.copy())
    xs = np.asarray(xs)
    assert np.all(np.abs(xs.mean(0) - mu) < 4 * sd / math.sqrt(200))
    assert np.all(np.abs(xs.std(0) / sd - 1) < 0.25)
    assert 0.5 < st["mean_tree_accept"] <= 1.0


def test_binary_gibbs_matches_bernoulli_target():
    rng = np.random.default_rng(2)
    p = np.array([0.1, 0.5, 0.9, 0.3])
    state = np.zeros(4, dtype=int)

    def logp():
        return float(np.sum(np.where(state == 1, np.log(p), np.log1p(-p))))

    def flip_logp(i):
        state[i] ^= 1
        return logp()

    def unflip(i):
        state[i] ^= 1

    lp = logp()
    acc = np.zeros(4)
    n = 6000
    for _ in range(n):
        lp, _, _ = binary_gibbs_sweep(4, flip_logp, unflip, lp, rng)
        assert abs(lp - logp()) < 1e-12
        acc += state
    assert np.all(np.abs(acc / n - p) < 0.03)


def test_cli_flags_match_reference():
    ps = build_parser()
    a = ps.parse_args(["--tune", "5", "--draws", "7"])
    assert (a.tune, a.draws, a.cores, a.ititers_data, a.netcdf) == (5, 7, None, "cohort_data", None)
    assert not a.split_delta and not a.split_omicron and not a.ignore_pcrpos
    a = ps.parse_args("--tune 1 --draws 2 --cores 4 --ititers_data d --split_delta --split_omicron --ignore_pcrpos --netcdf o.nc".split())
    assert (a.cores, a.ititers_data, a.netcdf) == (4, "d", "o.nc") and a.split_delta and a.split_omicron and a.ignore_pcrpos
    assert ps.prog == "abdpymc-infer"


def test_cli_thinning_flags_and_the_record_budget():
    """--thin / --record_every (thinned draws, posterior means always) and the host budget: at BASELINE config 3
    the default recording of 1000 draws x 4 chains would be 144 GB of host arrays -- refused with the thin that fits, before
    anything touches the device."""
    from types import SimpleNamespace

    from abdpymc_amd import sampler

    ps = build_parser()
    assert ps.parse_args("--tune 1 --draws 2".split()).thin == 1
    assert ps.parse_args("--tune 1 --draws 2 --thin 50".split()).thin == 50
    assert ps.parse_args("--tune 1 --draws 2 --record_every 7 --no_discrete".split()).thin == 7
    G, N = 200, 10000
    assert sampler.record_bytes(4, 1000, G, N, True, True) == 4 * 1000 * (G * N * 18 + N)  # 144 GB
    assert sampler.record_bytes(4, 1000, G, N, False, False) == 0
    fake = SimpleNamespace(ctx=None, n_gaps=G, n_inds=N)  # the check comes before the context is used
    with pytest.raises(ValueError, match=r"thin >= 1[78]"):
        sampler.sample_native(fake, 100, 1000, chains=4, budget_bytes=8 * 2 ** 30)
    with pytest.raises(ValueError, match="thin must be >= 1"):
        sampler.sample_native(fake, 100, 1000, chains=4, thin=0)
    # thin 50 at --draws 200 --chains 4: 4 draws per chain = 0.58 GB
    assert sampler.record_bytes(4, 4, G, N, True, True) < 2 ** 30
