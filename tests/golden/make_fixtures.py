#!/usr/bin/env python3
"""
Regenerates the committed fixtures under tests/golden/.  Runs ONLY in the build container (it reads
/root/reference); nothing at test / bench time reads the reference.

  test_cohort/            data files of the reference's own test cohort (data/test_data/cohort_data; GPL-3,
                          (c) the abdpymc authors) -- inputs, copied as data
  default_cohort.npz      the reference's default cohort (data/cohort_data) packed: the six arrays the model
                          needs (BASELINE config 1), as data
  loader_expect.json      what the REFERENCE's pure pandas/NumPy pieces return for those cohorts
                          (TiterData.from_disk sizes / index checksums, calculate_splits, logistic, and the
                          per-individual recurrence of simulation.py) -- obtained by importing abdpymc with inert
                          stand-ins for the modules that are not installed (arviz, pymc, pytensor, xarray);
                          no tensor code of the reference can run this way
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


def reference_loader_expectations():
    # inert stand-ins: the reference imports these at module top (abd.py:9-14) but its loader does not use them
    class _Inert(types.ModuleType):
        """any attribute (e.g. the annotation at.TensorLike) resolves to a placeholder; nothing is computed"""

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

    out = {}
    for key, sub in (("test_cohort", "data/test_data/cohort_data"), ("default_cohort", "data/cohort_data")):
        td = ref.TiterData.from_disk(os.path.join(REF, sub))
        e = dict(
            n_gaps=int(td.n_gaps), n_inds=int(td.n_inds), t0=str(td.t0),
            vacs_shape=list(td.vacs.shape), vacs_sum=float(td.vacs.sum()), pcrpos_sum=float(td.pcrpos.sum()),
            coords_ind=[int(td.coords["ind"][0]), int(td.coords["ind"][-1])],
            coords_gap=[int(td.coords["gap"][0]), int(td.coords["gap"][-1])],
            splits={f"{int(a)}{int(b)}": list(td.calculate_splits(delta=a, omicron=b)) for a in (False, True) for b in (False, True)},
        )
        for ag in ("s", "n"):
            a = getattr(td, ag)
            e[ag] = dict(
                n_obs=int(len(a.idx_gap)), n_gaps=int(a.n_gaps), n_inds=int(a.n_inds),
                idx_gap_sha=sha(np.asarray(a.idx_gap, dtype=np.int64)), idx_ind_sha=sha(np.asarray(a.idx_ind, dtype=np.int64)),
                log_dilution_sha=sha(a.df["log_dilution"].values.astype(np.float64)), od_sha=sha(a.df["od"].values.astype(np.float64)),
                od_sum=float(a.df["od"].values.sum()),
            )
        out[key] = e
    # logistic / invlogistic (abd.py:552-557)
    xs = [0.0, 1.0, 2.0, 4.0, 7.0]
    out["logistic"] = dict(a=1.3, b=-2.2, d=1.6, x=xs, y=[float(ref.logistic(x, a=1.3, b=-2.2, d=1.6)) for x in xs])
    # the reference's only pure-NumPy recurrence (simulation.py:117-132, 222-279), no randomness (lam0 = 0)
    ind = refsim.Individual(pcrpos=[0, 0, 1, 0, 0, 0, 1, 0], vacs=[0, 1, 0, 0, 0, 0, 0, 1])
    r = ind.infection_responses(lam0=np.zeros(8))
    out["simulation_recurrence"] = dict(
        pcrpos=[0, 0, 1, 0, 0, 0, 1, 0], vacs=[0, 1, 0, 0, 0, 0, 0, 1],
        init=-2.0, perm=2.0, temp_i=1.5, temp_v=2.0, wane=0.95,
        infections=[float(v) for v in r.infections], s=[float(v) for v in r.s_response], n=[float(v) for v in r.n_response],
    )
    json.dump(out, open(os.path.join(HERE, "loader_expect.json"), "w"), indent=1)


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
    reference_loader_expectations()
    oracle_logp_golden()
    for f in sorted(os.listdir(HERE)):
        p = os.path.join(HERE, f)
        if os.path.isfile(p):
            print(f, os.path.getsize(p))
