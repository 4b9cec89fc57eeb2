#!/bin/bash
# scheduler knobs of the lane-per-proposal sweep kernel (tuning build): refill threshold, tail
export ABD_HIP_LIB=$PWD/build/libabd_hip_tuning.so
for st in truth random; do
for rm in 4 8 12 16 24; do
  echo -n "$st refill_min=$rm: "; ABD_G2_REFILL_MIN=$rm python3 tools/probe_gibbs.py 20 $st 4
done
for tl in "4 6" "8 3" "8 12" "16 6" "0 6"; do
  set -- $tl
  echo -n "$st tail_lanes=$1 tail_age=$2: "; ABD_G2_TAIL_LANES=$1 ABD_G2_TAIL_AGE=$2 python3 tools/probe_gibbs.py 20 $st 4
done
done
