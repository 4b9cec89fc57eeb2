#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (one line per kernel).
usage: resource_usage.py build/libabd_hip.so.log  (written by __graft_entry__.build_hip whenever it links the library)"""
import re
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]

    def f(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    row = [
        name,
        f("TotalSGPRs"),
        f("VGPRs"),
        f("AGPRs"),
        f(r"ScratchSize \[bytes/lane\]"),
        f(r"Occupancy \[waves/SIMD\]"),
        f(r"LDS Size \[bytes/block\]"),
        f("SGPRs Spill"),
        f("VGPRs Spill"),
    ]
    print("{:52s} sgpr {:>4} vgpr {:>4} agpr {:>3} scratch {:>5} occ {:>2} lds {} spills sgpr {:>3} vgpr {:>3}".format(*row))
