"""
On-disk cohort format and the data container the model consumes.

Mirror of the reference's ``TiterData`` / ``AntigenTiterData`` (abdpymc/abd.py:22-221) restricted to what
the joint-logp path needs: the S / N observation lists with their (gap, ind) indexes, the (n_inds, n_gaps)
vaccination and PCR+ panels, sizes, coords and the variant split helper.  Same file layout
(``df.csv``, ``vacs.txt``, ``pcrpos.txt``, ``t0.txt``), same attribute names, same errors.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

# measurement codes that select the two antigens (abd.py:85, 94)
MEASUREMENT_S = "10222020-S"
MEASUREMENT_N = "40588-V08B"


class AntigenTiterData:
    """Titer data of one antigen (abd.py:22-43)."""

    def __init__(self, ag: str, idx_gap, idx_ind, log_dilution, od) -> None:
        self.ag = ag
        self.idx_gap = np.asarray(idx_gap, dtype=np.int64)  # df.elapsed_months   abd.py:35
        self.idx_ind = np.asarray(idx_ind, dtype=np.int64)  # df.individual_i     abd.py:36
        self.log_dilution = np.asarray(log_dilution, dtype=np.float64)  # abd.py:462
        self.od = np.asarray(od, dtype=np.float64)  # abd.py:468
        if not (self.idx_gap.shape == self.idx_ind.shape == self.log_dilution.shape == self.od.shape):
            raise ValueError("observation columns have different lengths")
        # sizes of the antigen's own sub-frame (abd.py:39-40); only informative
        self.n_gaps = int(self.idx_gap.max() + 1) if self.idx_gap.size else 0
        self.n_inds = int(self.idx_ind.max() + 1) if self.idx_ind.size else 0

    def __len__(self) -> int:
        return int(self.od.size)

    def __repr__(self) -> str:
        return f"AntigenTiterData(ag={self.ag}, n_obs={len(self)})"

    @property
    def obs(self):
        """(idx_gap, idx_ind, log_dilution, od) as the native Context takes them."""
        return self.idx_gap, self.idx_ind, self.log_dilution, self.od


def _month_index(period: str) -> int:
    """'YYYY-MM' -> months since year 0 (what pd.Period(freq='M') differences count)."""
    y, m = str(period).strip()[:7].split("-")
    return int(y) * 12 + int(m) - 1


class TiterData:
    """OD, PCR+ and vaccination data of a cohort (abd.py:46-221)."""

    def __init__(self, t0: str, s: AntigenTiterData, n: AntigenTiterData, vacs, pcrpos, n_gaps: int, n_inds: int,
                 record_ids=None, ageenroll=None) -> None:
        self.t0 = str(t0).strip()
        self.s = s
        self.n = n
        vacs = np.asarray(vacs)
        pcrpos = np.asarray(pcrpos)
        if vacs.shape != pcrpos.shape:
            raise ValueError("vacs and pcrpos are different shapes")  # abd.py:196-197
        # (n_inds, n_gaps) arrays with 1 where the event happened to an individual in a month (abd.py:112-115)
        self.vacs = vacs
        self.pcrpos = pcrpos
        self.n_gaps = int(n_gaps)  # max(df.elapsed_months) + 1   abd.py:101
        self.n_inds = int(n_inds)  # max(df.individual_i) + 1     abd.py:102
        self.coords = dict(ind=np.arange(self.n_inds), gap=np.arange(self.n_gaps))  # abd.py:126
        # record id of every individual in order of first appearance in the table (abd.py:104-110); not used by the model
        self.record_ids = None if record_ids is None else np.asarray(record_ids)
        if ageenroll is not None:  # optional enrollment ages keyed by record id (abd.py:128-131); not used by the model
            if self.record_ids is None:
                raise ValueError("ageenroll needs the table's record_id column")
            self.ageenroll = np.array([ageenroll[record_id] for record_id in self.record_ids])

    def __repr__(self) -> str:
        return f"TiterData(t0={self.t0}, n_inds={self.n_inds})"

    @classmethod
    def from_disk(cls, directory: str) -> "TiterData":
        """Read df.csv, vacs.txt, pcrpos.txt, t0.txt and, where present, individuals.csv (abd.py:171-202)."""
        import pandas as pd

        def path(x):
            return os.path.join(str(directory), x)

        df = pd.read_csv(path("df.csv"), index_col=0)
        vacs = np.loadtxt(path("vacs.txt"))
        pcrpos = np.loadtxt(path("pcrpos.txt"))
        try:  # enrollment ages, one "record_id,age" row per individual, no header (abd.py:189-194)
            ageenroll = pd.read_csv(path("individuals.csv"), header=None, index_col=0).squeeze("columns")
        except FileNotFoundError:
            ageenroll = None
        if vacs.shape != pcrpos.shape:
            raise ValueError("vacs and pcrpos are different shapes")
        with open(path("t0.txt"), "r") as fobj:
            t0 = fobj.readline().strip()
        return cls.from_frame(t0, df, vacs, pcrpos, ageenroll=ageenroll)

    @classmethod
    def from_frame(cls, t0: str, df, vacs, pcrpos, ageenroll=None) -> "TiterData":
        """Split the long table by measurement code (abd.py:82-98)."""

        def antigen(ag, code):
            sub = df[df["measurement"] == code]
            return AntigenTiterData(
                ag,
                sub["elapsed_months"].to_numpy(),
                sub["individual_i"].to_numpy(),
                sub["log_dilution"].to_numpy(),
                sub["od"].to_numpy(),
            )

        n_gaps = int(df["elapsed_months"].max()) + 1
        n_inds = int(df["individual_i"].max()) + 1
        record_ids = None
        if "record_id" in df.columns:
            record_ids = df[["individual_i", "record_id"]].drop_duplicates()["record_id"].to_numpy()
        return cls(t0, antigen("s", MEASUREMENT_S), antigen("n", MEASUREMENT_N), vacs, pcrpos, n_gaps, n_inds,
                   record_ids=record_ids, ageenroll=ageenroll)

    @classmethod
    def from_arrays(cls, n_gaps, n_inds, s_obs, n_obs, vacs, pcrpos, t0: str = "2020-05") -> "TiterData":
        """Build from in-memory arrays (synthetic cohorts, packed fixtures)."""
        return cls(t0, AntigenTiterData("s", *s_obs), AntigenTiterData("n", *n_obs), vacs, pcrpos, n_gaps, n_inds)

    def date_to_gap(self, period: str) -> int:
        """Which gap did a particular month occur in? (abd.py:143-147)"""
        return _month_index(period) - _month_index(self.t0)

    def calculate_splits(self, delta: bool, omicron: bool) -> tuple:
        """Gaps from t0 to when delta (2021-07) and / or omicron (2022-01) started to circulate (abd.py:204-221)."""
        splits = []
        if delta:
            splits.append(self.date_to_gap("2021-07"))
        if omicron:
            splits.append(self.date_to_gap("2022-01"))
        return tuple(splits)


def check_splits(splits, data: Optional[TiterData] = None) -> None:
    """Same conditions and messages as the reference (abd.py:604-622)."""
    if splits is not None:
        if any(split < 0 for split in splits):
            raise ValueError("split indexes must be positive")
        if sorted(splits) != list(splits):
            raise ValueError("splits must be in ascending order")
        if data is not None and splits and splits[-1] > data.n_gaps:
            raise ValueError(f"largest split must be less than n_gaps - 1, ({splits[-1]})")
        if len(splits) != len(set(splits)):
            raise ValueError("splits not unique")
        if any(not isinstance(split, int) for split in splits):
            raise ValueError("splits must be ints")
