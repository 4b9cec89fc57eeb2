#!/bin/bash
export ABD_PROBE_THETA_ROW=5 ABD_PROBE_SAME_STATE=1
for round in 1 2; do
for lib in abdpymc_amd/libabd_hip.so build/libabd_hip_onelevel0.so; do
  for c in 1 4; do echo -n "$lib chains=$c: "; ABD_HIP_LIB=$PWD/$lib python3 tools/probe_nuts_rate.py c3 $c 150 2>&1 | tail -1 | cut -c1-150; done
done
done
