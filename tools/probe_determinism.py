#!/usr/bin/env python3
"""Hunt for nondeterminism: the same short sampler run repeated many times must give identical bits."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abdpymc_amd.data import TiterData
from abdpymc_amd.model import model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gibbs = os.environ.get("PROBE_GIBBS", "1") == "1"
iters = int(os.environ.get("PROBE_ITERS", "28"))
td = TiterData.from_disk(os.path.join(ROOT, "tests", "golden", "test_cohort"))
m = model(td, n_chains=2)
pt = m.initial_point()
ref = None
bad = 0
for r in range(reps):
    q0 = np.empty((2, 17))
    for c in range(2):
        m.ctx.set_discrete(c, pt["i_raw"].astype(np.int8), pt["ab_s_waner"].astype(np.int8))
        q0[c] = m.ravel(pt) + 0.1 * np.random.default_rng([1, c]).uniform(-1, 1, 17)
    smp = m.ctx.sampler([0, 1], q0, tune=20, seed=1, gibbs=gibbs, accumulate=False)
    th, st = smp.run(iters)
    smp.close()
    if ref is None:
        ref, ref_st = th.copy(), st
    elif not np.array_equal(th, ref):
        bad += 1
        first = np.argwhere((th != ref).any(-1))
        c0, it0 = first[0].tolist()
        msg = [f"rep {r}: theta differs first at chain {c0}, iteration {it0}; chains affected {sorted(set(first[:, 0].tolist()))}"]
        for name in ("gibbs_accepted", "gibbs_proposed", "lp", "n_steps", "energy", "step_size"):
            d = np.argwhere(st[name][c0] != ref_st[name][c0]).ravel()
            msg.append(f"{name} first differs at {d[0] if d.size else None}")
        lo = max(0, it0 - 1)
        msg.append(f"lp[{lo}:{it0 + 1}] {st['lp'][c0, lo:it0 + 1]} vs {ref_st['lp'][c0, lo:it0 + 1]}; n_steps {st['n_steps'][c0, lo:it0 + 1]} vs {ref_st['n_steps'][c0, lo:it0 + 1]}; energy {st['energy'][c0, lo:it0 + 1]} vs {ref_st['energy'][c0, lo:it0 + 1]}")
        print("; ".join(msg), flush=True)
print(f"{bad} of {reps} runs differed (gibbs={gibbs}, OBS_LANES={os.environ.get('ABD_OBS_LANES', 'default')})")
