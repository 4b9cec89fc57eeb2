#!/usr/bin/env python3
"""
Regenerates the committed fixtures under tests/golden/.  Runs ONLY in the build container (it reads
/root/reference); nothing at test / bench time reads the reference.

  test_cohort/            data files of the reference's own test cohort (data/test_data/cohort_data; GPL-3,
                          (c) the abdpymc authors) -- inputs, copied as data
  default_cohort.npz      the reference's default cohort (data/cohort_data) packed: the six arrays the model
                          needs (BASELINE config 1), as data
  loader_expect.json      what a loader of the reference's on-disk format must return for those cohorts (sizes, index
                          checksums, splits), the logistic curve and the per-individual recurrence of simulation.py
                          -- derived from the DATA FILES and the documented rules with plain pandas / NumPy; no
                          reference code is imported.  --cross-check-reference-import additionally runs the
                          reference's own pure pandas/NumPy loader (inert stand-ins for its uninstalled imports)
                          and asserts it returns the same: a check, never a source of the committed values
  logp_golden.json        (theta, i_raw, waner) -> (logp, grad[17], checksums of i / mu_n / mu_s) for the test
                          cohort under all split combinations, from oracle/abd_oracle.py.  PARITY UNPINNED against
                          the reference itself: no reference test evaluates logp
"""
import hashlib
import json
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def copy_test_cohort():
    src = os.path.join(REF, "data", "test_data", "cohort_data")
    dst = os.path.join(HERE, "test_cohort")
    os.makedirs(dst, exist_ok=True)
    for f in ("df.csv", "vacs.txt", "pcrpos.txt", "t0.txt"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)


def pack_default_cohort():
    import pandas as pd

    d = os.path.join(REF, "data", "cohort_data")
    df = pd.read_csv(os.path.join(d, "df.csv"), index_col=0)
    vacs = np.loadtxt(os.path.join(d, "vacs.txt")).astype(np.int8)
    pcr = np.loadtxt(os.path.join(d, "pcrpos.txt")).astype(np.int8)
    t0 = open(os.path.join(d, "t0.txt")).readline().strip()
    is_s = (df["measurement"] == "10222020-S").to_numpy()
    is_n = (df["measurement"] == "40588-V08B").to_numpy()
    assert (is_s | is_n).all()
    np.savez_compressed(
        os.path.join(HERE, "default_cohort.npz"),
        is_s=is_s,
        elapsed_months=df["elapsed_months"].to_numpy().astype(np.int16),
        individual_i=df["individual_i"].to_numpy().astype(np.int16),
        log_dilution=df["log_dilution"].to_numpy().astype(np.float64),
        od=df["od"].to_numpy().astype(np.float64),
        vacs=vacs,
        pcrpos=pcr,
        t0=np.array(t0),
    )


def _months_between(t0: str, later: str) -> int:
    """(pd.Period(later) - pd.Period(t0)).n for monthly periods 'YYYY-MM', by calendar arithmetic"""
    y0, m0 = (int(v) for v in t0.split("-")[:2])
    y1, m1 = (int(v) for v in later.split("-")[:2])
    return (y1 - y0) * 12 + (m1 - m0)


def data_file_expectations():
    """
    What a loader of the reference's on-disk format must return, derived FROM THE DATA FILES THEMSELVES with plain
    pandas / NumPy (no reference code is imported): the rules are the documented ones of abd.py:22-43, 82-98, 171-221
    -- S rows are measurement '10222020-S', N rows '40588-V08B'; idx_gap = elapsed_months, idx_ind = individual_i;
    an antigen's n_gaps / n_inds are max + 1 of ITS rows; the cohort's sizes are the shape of vacs; splits are whole
    months from t0 to 2021-07 (delta) and 2022-01 (omicron).
    """
    import pandas as pd

    out = {}
    for key, d in (("test_cohort", os.path.join(REF, "data", "test_data", "cohort_data")),
                   ("default_cohort", os.path.join(REF, "data", "cohort_data"))):
        df = pd.read_csv(os.path.join(d, "df.csv"), index_col=0)
        vacs = np.loadtxt(os.path.join(d, "vacs.txt"))
        pcr = np.loadtxt(os.path.join(d, "pcrpos.txt"))
        t0 = open(os.path.join(d, "t0.txt")).readline().strip()
        assert vacs.shape == pcr.shape
        n_inds, n_gaps = vacs.shape
        e = dict(
            n_gaps=int(n_gaps), n_inds=int(n_inds), t0=t0[:7],
            vacs_shape=list(vacs.shape), vacs_sum=float(vacs.sum()), pcrpos_sum=float(pcr.sum()),
            coords_ind=[0, int(n_inds) - 1], coords_gap=[0, int(n_gaps) - 1],
            splits={f"{int(a)}{int(b)}": ([_months_between(t0, "2021-07")] if a else []) + ([_months_between(t0, "2022-01")] if b else [])
                    for a in (False, True) for b in (False, True)},
        )
        for ag, code in (("s", "10222020-S"), ("n", "40588-V08B")):
            sub = df[df["measurement"] == code]
            g, j = sub["elapsed_months"].to_numpy(), sub["individual_i"].to_numpy()
            e[ag] = dict(
                n_obs=int(len(sub)), n_gaps=int(g.max() + 1), n_inds=int(j.max() + 1),
                idx_gap_sha=sha(g.astype(np.int64)), idx_ind_sha=sha(j.astype(np.int64)),
                log_dilution_sha=sha(sub["log_dilution"].to_numpy().astype(np.float64)), od_sha=sha(sub["od"].to_numpy().astype(np.float64)),
                od_sum=float(sub["od"].to_numpy().sum()),
            )
        out[key] = e
    # logistic (abd.py:556-557): d / (1 + exp(-b (x - a))), evaluated here
    xs = [0.0, 1.0, 2.0, 4.0, 7.0]
    out["logistic"] = dict(a=1.3, b=-2.2, d=1.6, x=xs, y=[float(1.6 / (1.0 + np.exp(2.2 * (x - 1.3)))) for x in xs])
    # the per-individual recurrence of simulation.py:117-132, 222-279 with its defaults (init -2, perm 2, temp_i 1.5,
    # temp_v 2.0, wane 0.95; simulation.py:104-108), no randomness: response = init + perm [exposed] + temp, where
    # temp <- wane * temp + boost of the exposure (infection: temp_i on S and N; vaccination: temp_v on S only)
    pcrpos, vac = [0, 0, 1, 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 0, 0, 1]
    s_resp, n_resp, ts, tn, es, en = [], [], 0.0, 0.0, False, False
    for i, v in zip(pcrpos, vac):
        ts = 0.95 * ts + 1.5 * i + 2.0 * v
        tn = 0.95 * tn + 1.5 * i
        es, en = es or bool(i or v), en or bool(i)
        s_resp.append(-2.0 + (2.0 if es else 0.0) + ts)
        n_resp.append(-2.0 + (2.0 if en else 0.0) + tn)
    out["simulation_recurrence"] = dict(pcrpos=pcrpos, vacs=vac, init=-2.0, perm=2.0, temp_i=1.5, temp_v=2.0, wane=0.95,
                                        infections=[float(v) for v in pcrpos], s=s_resp, n=n_resp)
    return out


def reference_import_cross_check(expect: dict):
    """
    OPTIONAL (--cross-check-reference-import): the same quantities from the reference's own pure pandas / NumPy code,
    imported with inert stand-ins for the modules that are not installed (arviz, pymc, pytensor, xarray) -- no tensor
    code of the reference can run this way.  Only compares; the committed JSON never depends on it.
    """
    class _Inert(types.ModuleType):
        def __getattr__(self, key):
            if key.startswith("__"):
                raise AttributeError(key)
            return type(key, (), {})

    for name in ("arviz", "pymc", "pytensor", "pytensor.tensor", "xarray"):
        sys.modules.setdefault(name, _Inert(name))
    sys.modules["pytensor"].tensor = sys.modules["pytensor.tensor"]
    sys.path.insert(0, REF)
    import abdpymc as ref  # noqa: E402
    from abdpymc import simulation as refsim  # noqa: E402

    for key, sub in (("test_cohort", "data/test_data/cohort_data"), ("default_cohort", "data/cohort_data")):
        td = ref.TiterData.from_disk(os.path.join(REF, sub))
        e = expect[key]
        assert (int(td.n_gaps), int(td.n_inds), str(td.t0)) == (e["n_gaps"], e["n_inds"], e["t0"])
        for a in (False, True):
            for b in (False, True):
                assert list(td.calculate_splits(delta=a, omicron=b)) == e["splits"][f"{int(a)}{int(b)}"]
        for ag in ("s", "n"):
            x = getattr(td, ag)
            assert (int(len(x.idx_gap)), int(x.n_gaps), int(x.n_inds)) == (e[ag]["n_obs"], e[ag]["n_gaps"], e[ag]["n_inds"])
            assert sha(np.asarray(x.idx_gap, dtype=np.int64)) == e[ag]["idx_gap_sha"]
            assert sha(np.asarray(x.idx_ind, dtype=np.int64)) == e[ag]["idx_ind_sha"]
            assert sha(x.df["od"].values.astype(np.float64)) == e[ag]["od_sha"]
    for x, y in zip(expect["logistic"]["x"], expect["logistic"]["y"]):
        assert abs(float(ref.logistic(x, a=1.3, b=-2.2, d=1.6)) - y) <= 1e-15
    sr = expect["simulation_recurrence"]
    r = refsim.Individual(pcrpos=sr["pcrpos"], vacs=sr["vacs"]).infection_responses(lam0=np.zeros(8))
    assert np.allclose(r.s_response, sr["s"], rtol=0, atol=1e-12) and np.allclose(r.n_response, sr["n"], rtol=0, atol=1e-12)
    print("reference import cross-check: identical")


def oracle_logp_golden():
    from abdpymc_amd.data import TiterData
    from abdpymc_amd import synthetic
    from oracle import abd_oracle as O

    td = TiterData.from_disk(os.path.join(HERE, "test_cohort"))
    coh = O.Cohort(td.n_gaps, td.n_inds, td.vacs.astype(np.int8), td.pcrpos.astype(np.int8),
                   O.AntigenObs(*td.s.obs), O.AntigenObs(*td.n.obs))
    cases = []
    rng = np.random.default_rng(20240101)
    combos = [((), False), ((14,), False), ((20,), False), ((14, 20), False), ((), True), ((14, 20), True)]
    for splits, ignore in combos:
        for rep in range(2):
            theta = synthetic.theta_init(td.n_gaps) + 0.3 * rng.standard_normal(17)
            i_raw = (rng.random((td.n_gaps, td.n_inds)) < (0.08 if rep == 0 else 0.5)).astype(np.int8)
            w = (rng.random(td.n_inds) < 0.5).astype(np.int8)
            lp, g = O.logp_dlogp(theta, i_raw, w, coh, splits or None, ignore)
            lp_dense = O.joint_logp(theta, i_raw, w, coh, splits or None, ignore, dense=True)
            assert abs(lp - lp_dense) <= 1e-11 * abs(lp)
            i, mun, mus = O.deterministics(theta, i_raw, w, coh, splits or None, ignore)
            cases.append(dict(
                splits=list(splits), ignore_pcrpos=ignore, theta=theta.tolist(), i_raw=i_raw.tolist(), waner=w.tolist(),
                logp=lp, grad=g.tolist(), i=i.tolist(), mu_n_sum=float(mun.sum()), mu_s_sum=float(mus.sum()),
                mu_n_last=mun[-1].tolist(), mu_s_last=mus[-1].tolist(),
            ))
    json.dump(dict(_about="oracle/abd_oracle.py on tests/golden/test_cohort; parity unpinned vs the reference's logp",
                   cases=cases), open(os.path.join(HERE, "logp_golden.json"), "w"))


if __name__ == "__main__":
    copy_test_cohort()
    pack_default_cohort()
    expect = data_file_expectations()
    json.dump(expect, open(os.path.join(HERE, "loader_expect.json"), "w"), indent=1)
    if "--cross-check-reference-import" in sys.argv:
        reference_import_cross_check(expect)
    oracle_logp_golden()
    for f in sorted(os.listdir(HERE)):
        p = os.path.join(HERE, f)
        if os.path.isfile(p):
            print(f, os.path.getsize(p))
