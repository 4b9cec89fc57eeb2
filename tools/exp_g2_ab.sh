#!/bin/bash
# the sweep of the library in the tree against build/libabd_hip_prev.so: config 3, 1 and 4 chains, converged and random state;
# then the gap-count scaling probe of each
set -e
timeout -k 10 600 python3 -m pytest tests/test_gibbs.py -x -q
for lib in "" $PWD/build/libabd_hip_prev.so; do
  echo "== lib: ${lib:-tree}"
  for st in truth random; do
    for c in 1 4; do echo -n "$st $c chains: "; ABD_HIP_LIB=${lib:-$PWD/abdpymc_amd/libabd_hip.so} python3 tools/probe_gibbs.py 20 $st $c; done
  done
  ABD_HIP_LIB=${lib:-$PWD/abdpymc_amd/libabd_hip.so} timeout -k 10 300 python3 tools/probe_gibbs_gaps.py
done
