#!/bin/bash
# Experiment (round 4): what do dependent launch chains of 1 / 2 / 4 chains per launch deliver when 4 / 2 / 1 of them share
# the chip, with and without the split panels?  Stream-ordered launches on one stream stand in for a leapfrog train.
export ABD_HIP_LIB=$PWD/build/libabd_hip_tuning.so
run() {  # label, chains, pipes, xc_max_cb
  ABD_PIPES=$3 ABD_XC_MAX_CB=$4 python3 bench.py --no-cpu-baseline --no-sampler --steps 200 --chains $2 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ro=d['roofline']
print('$1: chains/launch $2, streams $3, xc<=$4: %.0f evals/s, %.2f us per launch (timed shape), %.2f isolated, sync %.0f' % (d['value'], ro['kernel_us'], ro['isolated']['kernel_us'], d['sync_evals_per_s']))"
}
for round in 1 2; do
run "1x4" 4 1 1
run "1x4xc" 4 1 4
run "2x2" 2 2 1
run "2x2xc" 2 2 4
run "4x1xc" 1 4 1
run "4x4 (headline shape)" 4 4 1
run "4x4xc" 4 4 4
done
