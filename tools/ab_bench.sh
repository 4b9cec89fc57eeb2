#!/bin/bash
# A/B two builds of the library on ONE box through bench.py's stream-ordered region (timings from different gpurun boxes
# differ by up to ~10 %).  usage: tools/ab_bench.sh build/libA.so build/libB.so [bench args...]
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for lib in $A $B; do
    ABD_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-sampler "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ro=d['roofline']
print('$lib round $round: %.0f evals/s, %.2f us per launch (timed shape), %.2f isolated, sync %.0f' % (d['value'], ro['kernel_us'], ro['isolated']['kernel_us'], d['sync_evals_per_s']))"
  done
done
