#!/bin/bash
# NUTS-seen rate at config 3 over the train units' shape (chains per unit) and look-ahead, one box
export ABD_PROBE_THETA_ROW=5 ABD_HIP_LIB=$PWD/build/libabd_hip_tuning.so
IT=${IT:-150}
for same in 1 0; do
for unit in 1 2 4; do
  for la in 2 3 5; do
    echo -n "same_state=$same unit=$unit lookahead=$la: "
    ABD_PROBE_SAME_STATE=$same ABD_SAMPLER_UNIT=$unit ABD_TRAIN_LOOKAHEAD=$la python3 tools/probe_nuts_rate.py c3 4 $IT 2>&1 | tail -1
  done
done
done
