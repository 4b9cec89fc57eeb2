#!/bin/bash
# the compound sampler end to end at config 3 (4 chains, NUTS + sweep, nothing recorded): round 3's tree against this one, one box
for round in 1 2; do
  echo -n "r03: "; (cd build/r03tree && python3 tools/bench_sampler.py c3 --tune 100 --draws 100 --no-record 2>&1 | tail -1)
  for unit in 1 2; do
    echo -n "now unit=$unit: "; ABD_SAMPLER_UNIT=$unit python3 tools/bench_sampler.py c3 --tune 100 --draws 100 --no-record 2>&1 | tail -1
  done
done
