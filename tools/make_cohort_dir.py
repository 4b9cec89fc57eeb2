#!/usr/bin/env python3
"""Write a synthetic dense cohort (abdpymc_amd.synthetic.make_cohort: BASELINE configs 2-5) as a cohort DIRECTORY in the
reference's on-disk format (abd.py:171-202: df.csv, vacs.txt, pcrpos.txt, t0.txt), for abdpymc-infer --ititers_data.
usage: make_cohort_dir.py DIR [n_inds n_gaps]   (default 10000 200: 4 M table rows, ~190 MB)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd import synthetic  # noqa: E402
from abdpymc_amd.data import MEASUREMENT_N, MEASUREMENT_S  # noqa: E402


def write_cohort_dir(path: str, n_inds: int = 10000, n_gaps: int = 200) -> None:
    import pandas as pd

    sc = synthetic.make_cohort(n_inds, n_gaps)
    os.makedirs(path, exist_ok=True)
    k = sc.idx_gap.size
    df = pd.DataFrame({
        "measurement": np.concatenate([np.full(k, MEASUREMENT_S), np.full(k, MEASUREMENT_N)]),
        "od": np.concatenate([sc.y_s, sc.y_n]),
        "elapsed_months": np.concatenate([sc.idx_gap, sc.idx_gap]),
        "individual_i": np.concatenate([sc.idx_ind, sc.idx_ind]),
        "log_dilution": np.concatenate([sc.x_s, sc.x_n]),
    })
    df.to_csv(os.path.join(path, "df.csv"), float_format="%.17g")
    np.savetxt(os.path.join(path, "vacs.txt"), sc.vacs, fmt="%d")
    np.savetxt(os.path.join(path, "pcrpos.txt"), sc.pcrpos, fmt="%d")
    with open(os.path.join(path, "t0.txt"), "w") as f:
        f.write("2020-05\n")


if __name__ == "__main__":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    g = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    write_cohort_dir(sys.argv[1], n, g)
    print(f"wrote {sys.argv[1]}: {n} individuals x {g} gaps")
