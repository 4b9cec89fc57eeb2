#!/bin/bash
# NUTS-seen rate at config 3 over the train units' shape (chains per unit), one box
export ABD_PROBE_THETA_ROW=5 ABD_HIP_LIB=$PWD/build/libabd_hip_tuning.so ABD_PROBE_SAME_STATE=1
IT=${IT:-100}
for cu in "4 1" "4 2" "8 1" "8 2" "8 4" "2 1" "2 2" "1 1" "16 2" "16 4"; do
  set -- $cu
  echo -n "chains=$1 unit=$2: "
  ABD_SAMPLER_UNIT=$2 python3 tools/probe_nuts_rate.py c3 $1 $IT 2>&1 | tail -1
done
