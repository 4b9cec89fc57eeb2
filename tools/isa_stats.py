#!/usr/bin/env python3
"""Instruction counts per kernel of a device-only assembly listing:
   hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S abdpymc_amd/csrc/abd_eval.hip -o build/eval.s
   python tools/isa_stats.py build/abd.s [name filter]
Used to check that a refactoring leaves a tuned kernel's code unchanged (and to count what a change costs)."""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = [l.strip().split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    kinds = {}
    for i in ins:
        k = i.split("_")[0] if not i.startswith(("global", "buffer", "ds", "flat", "scratch")) else i.split("_")[0]
        kinds[k] = kinds.get(k, 0) + 1
    vg = re.search(r"\.vgpr_count:\s+(\d+)", txt[txt.find(".name:           " + name):][:3000] or "")
    print(f"{name[:64]:64s} total {len(ins):5d} " + " ".join(f"{k} {v}" for k, v in sorted(kinds.items())))
