#!/usr/bin/env python3
"""Rewrite the counter-derived numbers of profiles/traffic.json from the PMC summaries of one tools/profile_round.sh run:
usage: tools/update_traffic.py profiles/r03 PREFIX   (PREFIX = the run's file prefix, e.g. "d_")
Reads PREFIX{c3,c5,c2}_pmc_sq.txt / _pmc_fetch_size.txt / _pmc_write_size_l2.txt (tools/pmc_summary.py output: one block
per (kernel, grid)) and PREFIXkernel_sources_sha256.txt; keeps the notes and the algorithmic byte counts."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, pre = sys.argv[1], sys.argv[2]
path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(path))


def blocks(fn):
    """{workgroups: {counter: mean}} of the dense kernel's blocks in a pmc_summary file"""
    out, cur = {}, None
    for line in open(os.path.join(ROOT, d, fn)):
        m = re.search(r"abd_dense_kernel.*= (\d+) workgroups", line)
        if m:
            cur = out.setdefault(int(m.group(1)), {})
            continue
        if line.startswith("void ") or line.startswith("abd_"):
            cur = None
            continue
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=([0-9.e+]+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def hbm(f, w):
    return int(round(f * 1024 * 2 + w * 1024))


rel = os.path.relpath(os.path.join(ROOT, d), ROOT)
for cfg in ("c3", "c5", "c2"):
    sq, fe, wr = blocks(f"{pre}{cfg}_pmc_sq.txt"), blocks(f"{pre}{cfg}_pmc_fetch_size.txt"), blocks(f"{pre}{cfg}_pmc_write_size_l2.txt")
    grids = sorted(sq)
    full, pipe = grids[-1], grids[0]
    e = t[cfg]
    e.update(fetch_size_kib_raw=fe[full]["FETCH_SIZE"], write_size_kib_raw=wr[full]["WRITE_SIZE"],
             hbm_bytes_per_launch=hbm(fe[full]["FETCH_SIZE"], wr[full]["WRITE_SIZE"]),
             valu_insts_per_launch=int(sq[full]["SQ_INSTS_VALU"]),
             valu_source=f"{rel}/{pre}{cfg}_pmc_sq.txt (SQ_INSTS_VALU, {full} workgroups)")
    if "pipe_grid" in e and pipe != full:
        e["pipe_grid"].update(workgroups=pipe, fetch_size_kib_raw=fe[pipe]["FETCH_SIZE"], write_size_kib_raw=wr[pipe]["WRITE_SIZE"],
                              hbm_bytes_per_launch=hbm(fe[pipe]["FETCH_SIZE"], wr[pipe]["WRITE_SIZE"]),
                              valu_insts_per_launch=int(sq[pipe]["SQ_INSTS_VALU"]),
                              valu_source=f"{rel}/{pre}{cfg}_pmc_sq.txt (SQ_INSTS_VALU, {pipe} workgroups)")
    print(cfg, "full grid", full, int(sq[full]["SQ_INSTS_VALU"]), e["hbm_bytes_per_launch"], "pipe grid", pipe, int(sq[pipe]["SQ_INSTS_VALU"]))
t["kernel_sources_sha256"] = open(os.path.join(ROOT, d, f"{pre}kernel_sources_sha256.txt")).read().strip()
t["_about"] = re.sub(r"profiles/r\d+/\w_\*_pmc_\*\.txt", f"{rel}/{pre}*_pmc_*.txt", t["_about"])
json.dump(t, open(path, "w"), indent=1)
print("kernel_sources_sha256", t["kernel_sources_sha256"])
