#!/bin/bash
# The A/B experiments of round 4 that profiles/README.md quotes, each on ONE GPU box (boxes of the pool differ by ~10 %).
# usage: tools/experiments.sh NAME     (needs the tuning build unless noted: python tools/build_variant.py tuning)
#   launch_shapes  dependent launch chains of 1 / 2 / 4 chains per launch, 4 / 2 / 1 of them sharing the chip, split panels
#                  or not: the ceiling a sampler with four evaluations in flight can reach (stream-ordered launches on one
#                  stream stand in for a leapfrog train)
#   train_units    NUTS-seen rate at config 3 over chains and chains per train unit (ABD_SAMPLER_UNIT)
#   train_grid     workgroups per CU of a one-chain train launch, 1 / 2 / 4 chains
#   handoff        the own-sum hand-off as shipped against the release/acquire forms (-DABD_HANDOFF_FORMAL=1 / 2):
#                  python tools/build_variant.py formal1 -DABD_HANDOFF_FORMAL=1; ... formal2 -DABD_HANDOFF_FORMAL=2
#   sync_own_sum   a synchronous call's launch summing its own rows against the second launch (ABD_SYNC_OWN_SUM)
#   sweep_knobs    scheduler constants of the lane-per-proposal sweep kernel
#   sweep_ab       the sweep of the library in the tree against another build (LIB_B=path; product libraries, no tuning build)
set -e
T=$PWD/build/libabd_hip_tuning.so
P=$PWD/abdpymc_amd/libabd_hip.so
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ro=d['roofline']
print('%.0f evals/s, %.2f us per launch (timed shape), %.2f isolated, sync %.0f' % (d['value'], ro['kernel_us'], ro['isolated']['kernel_us'], d['sync_evals_per_s']))"; }
nuts() { python3 tools/probe_nuts_rate.py "$@" 2>&1 | tail -1 | cut -c1-190; }
export ABD_PROBE_THETA_ROW=5 ABD_PROBE_SAME_STATE=1
case "$1" in
launch_shapes)
  export ABD_HIP_LIB=$T
  for s in "1x4 4 1 1" "1x4xc 4 1 4" "2x2 2 2 1" "2x2xc 2 2 4" "4x1xc 1 4 1" "4x4(headline) 4 4 1" "4x4xc 4 4 4"; do
    set -- $s
    echo -n "$1: chains/launch $2, streams $3, split panels for <= $4 chains: "
    ABD_PIPES=$3 ABD_XC_MAX_CB=$4 python3 bench.py --no-cpu-baseline --no-sampler --no-other-configs --steps 200 --chains $2 | line
  done ;;
train_units)
  export ABD_HIP_LIB=$T
  for cu in "1 1" "2 1" "2 2" "4 1" "4 2" "4 4" "8 1" "8 2" "8 4" "16 2" "16 4"; do
    set -- $cu
    echo -n "chains=$1 unit=$2: "; ABD_SAMPLER_UNIT=$2 nuts c3 $1 ${IT:-100}
  done ;;
train_grid)
  export ABD_HIP_LIB=$T ABD_SAMPLER_UNIT=1
  for b in 1 2 4; do for c in 1 2 4; do
    echo -n "workgroups/CU=$b chains=$c: "; ABD_TRAIN_BLOCKS_PER_CU=$b nuts c3 $c 100
  done; done ;;
handoff)
  for lib in $P $PWD/build/libabd_hip_formal1.so $PWD/build/libabd_hip_formal2.so; do
    echo -n "$(basename $lib) c3: "; ABD_HIP_LIB=$lib nuts c3 4 200
    echo -n "$(basename $lib) default cohort: "; ABD_HIP_LIB=$lib nuts default 4 300
    echo -n "$(basename $lib) bench: "; ABD_HIP_LIB=$lib python3 bench.py --no-cpu-baseline --no-sampler --no-other-configs --steps 200 | line
  done ;;
sync_own_sum)
  export ABD_HIP_LIB=$T
  for v in 0 1; do echo "ABD_SYNC_OWN_SUM=$v"; ABD_SYNC_OWN_SUM=$v python3 tools/probe_sync_latency.py; done ;;
sweep_knobs)
  export ABD_HIP_LIB=$T
  for st in truth random; do
    for rm in 4 8 12 16 24; do echo -n "$st refill_min=$rm: "; ABD_G2_REFILL_MIN=$rm python3 tools/probe_gibbs.py 20 $st 4; done
    for tl in "4 6" "8 3" "8 12" "16 6" "0 6"; do
      set -- $tl
      echo -n "$st tail_lanes=$1 tail_age=$2: "; ABD_G2_TAIL_LANES=$1 ABD_G2_TAIL_AGE=$2 python3 tools/probe_gibbs.py 20 $st 4
    done
  done ;;
sweep_ab)
  timeout -k 10 600 python3 -m pytest tests/test_gibbs.py -x -q
  for lib in $P ${LIB_B:?LIB_B=the other library}; do
    echo "== $lib"
    for st in truth random; do for c in 1 4; do echo -n "$st, $c chains: "; ABD_HIP_LIB=$lib python3 tools/probe_gibbs.py 20 $st $c; done; done
    ABD_HIP_LIB=$lib timeout -k 10 300 python3 tools/probe_gibbs_gaps.py
  done ;;
*) sed -n 2,16p "$0"; exit 2 ;;
esac
