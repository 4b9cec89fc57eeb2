#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (one line per kernel)."""
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
    ]
    print("{:52s} sgpr {:>4} vgpr {:>4} agpr {:>3} scratch {:>5} occ {:>2} lds {}".format(*row))
