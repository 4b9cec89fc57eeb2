"""Shared helpers for tests: build oracle Cohort objects from synthetic / on-disk cohorts."""
import numpy as np

from oracle import abd_oracle as O


def oracle_cohort_from_synth(sc) -> O.Cohort:
    return O.Cohort(
        n_gaps=sc.n_gaps,
        n_inds=sc.n_inds,
        vacs=sc.vacs,
        pcrpos=sc.pcrpos,
        s=O.AntigenObs(sc.idx_gap, sc.idx_ind, sc.x_s, sc.y_s),
        n=O.AntigenObs(sc.idx_gap, sc.idx_ind, sc.x_n, sc.y_n),
    )


def random_sparse_cohort(n_inds, n_gaps, k_s, k_n, seed=0):
    """Sparse observation lists with repeats per cell (several dilutions of one serum sample)."""
    rng = np.random.default_rng(seed)
    vacs = (rng.random((n_inds, n_gaps)) < 1.5 / n_gaps).astype(np.int8)
    pcr = (rng.random((n_inds, n_gaps)) < 1.0 / n_gaps).astype(np.int8)

    def obs(k):
        g = rng.integers(0, n_gaps, size=k).astype(np.int32)
        j = rng.integers(0, n_inds, size=k).astype(np.int32)
        x = rng.choice(np.array([0.0, 1.0, 2.0, 4.0, 6.0]), size=k)
        y = rng.uniform(-0.05, 2.1, size=k)
        return O.AntigenObs(g, j, x, y)

    return O.Cohort(n_gaps, n_inds, vacs, pcr, obs(k_s), obs(k_n))


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
