#!/bin/bash
# Collect the evidence bench.py's numbers rest on, on ONE GPU box: usage tools/profile_round.sh OUTDIR [configs...]
# (run from the repo root; writes under OUTDIR, which should be below gpurun_out/).  Every rocprofv3 command puts
# the program itself after "--" (python3 <script>); counters are collected in their own passes.
set -u
OUT=${1:-gpurun_out/prof}; shift || true
CFGS=${*:-c3}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
summ() { python3 tools/pmc_summary.py "$1" abd_ > "$2" 2>&1; }
for cfg in $CFGS; do
  case $cfg in
    c3) SW="--n-inds 10000 --n-gaps 200 --chains 4 --storage f64 --cpw 4 --blocks 1024,256" ;;
    c2) SW="--n-inds 1000 --n-gaps 60 --chains 4 --storage f64 --cpw 4 --blocks 0" ;;
    c5) SW="--n-inds 100000 --n-gaps 200 --chains 1 --storage f32 --cpw 1 --blocks 0,256" ;;
    c1) SW="" ;;
  esac
  echo "== $cfg: bench line" ; python3 bench.py --config $cfg > "$OUT/${cfg}_bench.json" 2> "$OUT/${cfg}_bench.err" || { echo bench failed; tail -5 "$OUT/${cfg}_bench.err"; exit 1; }
  echo "== $cfg: kernel trace of the same command"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_trace" -- python3 bench.py --config $cfg --no-cpu-baseline > "$OUT/${cfg}_bench_under_rocprof.json" 2> "$OUT/${cfg}_trace.err" || { echo trace failed; tail -5 "$OUT/${cfg}_trace.err"; exit 1; }
  find "$OUT/${cfg}_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${cfg}_kernel_stats.csv" \;
  echo "== $cfg: isolated kernel (ABD_PIPES=1)"
  ABD_PIPES=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_trace1" -- python3 bench.py --config $cfg --no-cpu-baseline > "$OUT/${cfg}_bench_one_pipe_under_rocprof.json" 2> "$OUT/${cfg}_trace1.err" || { echo trace1 failed; exit 1; }
  find "$OUT/${cfg}_trace1" -name "*kernel_stats.csv" -exec cp {} "$OUT/${cfg}_one_pipe_kernel_stats.csv" \;
  [ -z "$SW" ] && { rm -rf "$OUT/${cfg}_trace" "$OUT/${cfg}_trace1"; continue; }  # observation lists: launch-bound, no counter passes
  echo "== $cfg: PMC passes over tools/sweep.py $SW"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT/${cfg}_pmc_sq" --output-format csv -- python3 tools/sweep.py $SW --iters 10 > "$OUT/${cfg}_pmc_sq.log" 2>&1 || { echo pmc sq failed; tail -5 "$OUT/${cfg}_pmc_sq.log"; exit 1; }
  summ "$OUT/${cfg}_pmc_sq" "$OUT/${cfg}_pmc_sq.txt"
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/${cfg}_pmc_fetch" --output-format csv -- python3 tools/sweep.py $SW --iters 10 > "$OUT/${cfg}_pmc_fetch.log" 2>&1 || { echo pmc fetch failed; exit 1; }
  summ "$OUT/${cfg}_pmc_fetch" "$OUT/${cfg}_pmc_fetch_size.txt"
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/${cfg}_pmc_write" --output-format csv -- python3 tools/sweep.py $SW --iters 10 > "$OUT/${cfg}_pmc_write.log" 2>&1 || { echo pmc write failed; exit 1; }
  summ "$OUT/${cfg}_pmc_write" "$OUT/${cfg}_pmc_write_size_l2.txt"
  rm -rf "$OUT/${cfg}_trace" "$OUT/${cfg}_trace1" "$OUT/${cfg}_pmc_sq" "$OUT/${cfg}_pmc_fetch" "$OUT/${cfg}_pmc_write"
done
ls -la "$OUT"
# ---- the Gibbs sweep (config 3, 4 chains): time, scheduler statistics, SQ counters; random state and converged state ----
if [ "${GIBBS:-1}" = "1" ]; then
  for mode in random truth; do
    arg=""; [ $mode = truth ] && arg="truth"
    python3 tools/probe_gibbs.py 10 $arg > "$OUT/gibbs_${mode}_time.txt" 2>&1
    ABD_GIBBS_STATS=1 python3 tools/probe_gibbs.py 2 $arg > "$OUT/gibbs_${mode}_stats.txt" 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT/gibbs_pmc" --output-format csv -- python3 tools/probe_gibbs.py 3 $arg > "$OUT/gibbs_pmc.log" 2>&1
    python3 tools/pmc_summary.py "$OUT/gibbs_pmc" gibbs > "$OUT/gibbs_${mode}_pmc_sq.txt" 2>&1
    rm -rf "$OUT/gibbs_pmc"
    ABD_GIBBS_V1=1 python3 tools/probe_gibbs.py 10 $arg > "$OUT/gibbs_${mode}_time_wave_per_proposal_kernel.txt" 2>&1
  done
  for c in 1 2 4 8 16; do python3 tools/probe_nuts_rate.py c3 $c 300; done > "$OUT/nuts_rate_c3.txt" 2>&1
  ABD_SAMPLER_TRAINS=0 python3 tools/probe_nuts_rate.py c3 4 300 > "$OUT/nuts_rate_c3_without_trains.txt" 2>&1
  python3 tools/probe_sync_latency.py > "$OUT/sync_latency_c3.txt" 2>&1
  for c in 4 16; do python3 tools/probe_nuts_rate.py default $c 300; done > "$OUT/nuts_rate_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py default --chains 4 --tune 1000 --draws 1000 > "$OUT/sampler_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py default --chains 4 --tune 1000 --draws 1000 --dense >> "$OUT/sampler_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py default --chains 16 --tune 1000 --draws 1000 --no-record >> "$OUT/sampler_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py c3 --chains 4 --tune 100 --draws 100 --no-record > "$OUT/sampler_c3.txt" 2>&1
fi
python3 -c "import bench; print(bench.kernel_sources_sha256())" > "$OUT/kernel_sources_sha256.txt"
ls "$OUT" | wc -l
