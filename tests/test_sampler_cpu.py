"""The sampler machinery on closed-form targets (no GPU): NUTS on a correlated-scale Gaussian, binary Gibbs on
independent Bernoullis, and the CLI's flag set (reference abd.py:888-911)."""
import math

import numpy as np

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
