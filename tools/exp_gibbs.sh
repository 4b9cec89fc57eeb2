#!/bin/bash
# one-chain and four-chain sweeps at config 3, converged ("truth") and random state, over the waves per CU of a one-chain sweep
export ABD_HIP_LIB=$PWD/build/libabd_hip_tuning.so
for st in truth random; do
  for w in 8 12; do echo -n "$st 1 chain, $w waves/CU: "; ABD_G2_WAVES_ONE=$w python3 tools/probe_gibbs.py 20 $st 1; done
  echo -n "$st 2 chains: "; python3 tools/probe_gibbs.py 20 $st 2
  echo -n "$st 4 chains: "; python3 tools/probe_gibbs.py 20 $st 4
  echo -n "$st 8 chains: "; python3 tools/probe_gibbs.py 10 $st 8
done
ABD_GIBBS_STATS=1 python3 tools/probe_gibbs.py 1 truth 1 2>&1 | tail -2
