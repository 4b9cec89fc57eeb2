#!/bin/bash
# A/B of the own-sum hand-off: release/acquire as the memory model defines it (the product) against the drained-stores form
export ABD_PROBE_THETA_ROW=5 ABD_PROBE_SAME_STATE=1
for round in 1; do
  for lib in build/libabd_hip_formal2.so build/libabd_hip_relaxed.so; do
    echo -n "lib=$lib c3: "; ABD_HIP_LIB=$PWD/$lib python3 tools/probe_nuts_rate.py c3 4 200 2>&1 | tail -1 | cut -c1-160
    echo -n "lib=$lib default cohort: "; ABD_HIP_LIB=$PWD/$lib python3 tools/probe_nuts_rate.py default 4 300 2>&1 | tail -1 | cut -c1-160
    echo -n "lib=$lib bench: "; ABD_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-sampler --steps 200 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f evals/s, sync %.0f' % (d['value'], d['sync_evals_per_s']))"
  done
done
