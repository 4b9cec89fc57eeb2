#!/bin/bash
# workgroups per CU of a one-chain train launch: the rate of 1, 2 and 4 chains at config 3 (tuning build)
export ABD_PROBE_THETA_ROW=5 ABD_HIP_LIB=$PWD/build/libabd_hip_tuning.so ABD_PROBE_SAME_STATE=1 ABD_SAMPLER_UNIT=1
for b in 1 2 4; do
  for c in 1 2 4; do
    echo -n "blocks/CU=$b chains=$c: "; ABD_TRAIN_BLOCKS_PER_CU=$b python3 tools/probe_nuts_rate.py c3 $c 100 2>&1 | tail -1 | cut -c1-170
  done
done
