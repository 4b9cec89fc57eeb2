#!/bin/bash
# N processes x C chains each on one GPU, default cohort
N=$1; C=$2
t0=$(date +%s.%N)
pids=()
for i in $(seq 1 $N); do
  python tools/bench_sampler.py default --chains $C --tune 1000 --draws 1000 --no-record > gpurun_out/mp_$i.txt 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
t1=$(date +%s.%N)
echo "$N processes x $C chains: wall $(echo "$t1 - $t0" | bc) s (includes start-up)"
tail -qn1 gpurun_out/mp_*.txt | cut -c1-90
rm -f gpurun_out/mp_*.txt
