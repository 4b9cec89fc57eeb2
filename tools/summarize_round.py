#!/usr/bin/env python3
"""Key numbers of one tools/profile_round.sh run, for profiles/README.md: usage tools/summarize_round.py DIR [PREFIX]"""
import json
import os
import re
import sys

d = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else ""


def first_line(fn, n=1):
    p = os.path.join(d, pre + fn)
    return [ln.rstrip() for ln in open(p)][:n] if os.path.exists(p) else []


for c in ("c3", "c5", "c2", "c1"):
    p = os.path.join(d, f"{pre}{c}_bench.json")
    if not os.path.exists(p):
        continue
    b = json.load(open(p))
    ro, cs = b["roofline"], b["compound_step"]
    print(f"{c}: {b['value']:.0f} evals/s ({b['ms_per_step'] * 1e3:.2f} us per step), device {ro['kernel_us']:.2f} us timed / "
          f"{ro['isolated']['kernel_us']:.2f} isolated, HBM frac {ro['frac']:.3f} / {ro['isolated']['frac']:.3f}, "
          f"valu {ro['valu']['frac'] if ro.get('valu') else None}, sync {b['sync_evals_per_s']:.0f}, NUTS {b['nuts_evals_per_s']:.0f} "
          f"{b['nuts']['leapfrogs_per_chain']}, sweep {cs['gibbs_sweep_ms']} / {cs['gibbs_sweep_ms_converged_state']} ms, "
          f"cpu {b['cpu_baseline']['value']:.0f} ({b['cpu_baseline']['cores']} threads) / {b['cpu_baseline']['one_thread']['value']:.1f} (1)")
for fn in ("nuts_rate_c3.txt", "nuts_rate_c3_without_trains.txt", "nuts_rate_default_cohort.txt", "sync_latency_c3.txt", "sampler_c3.txt",
           "sampler_default_cohort.txt", "gibbs_random_time.txt", "gibbs_truth_time.txt"):
    for ln in first_line(fn, 8):
        m = re.match(r"(\S+ chains=\d+ \S+: [\d,]+ evals/s)|(.*per synchronous call.*)|(.*chain-iterations/s)|(.*sweep [\d.]+ ms)", ln)
        if m:
            print(f"{fn}: {ln[:110]}")
for c in ("c3", "c1"):
    for ln in first_line(f"{c}_one_pipe_kernel_stats.csv", 3)[1:]:
        print(f"{c} one pipe: {ln[:120]}")
