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
  echo "== $cfg: bench line" ; python3 bench.py --config $cfg --no-other-configs > "$OUT/${cfg}_bench.json" 2> "$OUT/${cfg}_bench.err" || { echo bench failed; tail -5 "$OUT/${cfg}_bench.err"; exit 1; }
  echo "== $cfg: kernel trace of the same command"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_trace" -- python3 bench.py --config $cfg --no-cpu-baseline --no-other-configs > "$OUT/${cfg}_bench_under_rocprof.json" 2> "$OUT/${cfg}_trace.err" || { echo trace failed; tail -5 "$OUT/${cfg}_trace.err"; exit 1; }
  find "$OUT/${cfg}_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${cfg}_kernel_stats.csv" \;
  if [ $cfg = c3 ]; then
    echo "== the driver's command: python3 bench.py --gpus 1 --steps 20 --warmup 5 (every single-GPU BASELINE config in one line)"
    ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/driver_bench.json" ) 2> "$OUT/driver_bench.err" || { echo driver bench failed; tail -5 "$OUT/driver_bench.err"; exit 1; }
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/driver_trace" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/driver_bench_under_rocprof.json" 2> "$OUT/driver_trace.err" || { echo driver trace failed; exit 1; }
    find "$OUT/driver_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/driver_kernel_stats.csv" \;
    rm -rf "$OUT/driver_trace"
  fi
  echo "== $cfg: isolated kernel (ABD_PIPES=1)"
  ABD_PIPES=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_trace1" -- python3 bench.py --config $cfg --no-cpu-baseline --no-sampler > "$OUT/${cfg}_bench_one_pipe_under_rocprof.json" 2> "$OUT/${cfg}_trace1.err" || { echo trace1 failed; exit 1; }
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
  python3 tools/probe_gibbs_gaps.py 200 256 300 512 > "$OUT/gibbs_gaps.txt" 2>&1
  ABD_GIBBS_V1=1 python3 tools/probe_gibbs_gaps.py 200 300 >> "$OUT/gibbs_gaps.txt" 2>&1
  python3 tools/probe_gibbs_scaling.py > "$OUT/gibbs_one_chain_by_cohort_size.txt" 2>&1
  for c in 1 2 4 8 16; do ABD_PROBE_SAME_STATE=1 ABD_PROBE_THETA_ROW=5 python3 tools/probe_nuts_rate.py c3 $c 200; done > "$OUT/nuts_rate_c3.txt" 2>&1
  for u in 1 2 4; do ABD_SAMPLER_UNIT=$u ABD_PROBE_SAME_STATE=1 ABD_PROBE_THETA_ROW=5 python3 tools/probe_nuts_rate.py c3 8 200; done > "$OUT/nuts_rate_c3_8_chains_by_unit.txt" 2>&1
  ABD_SAMPLER_TRAINS=0 ABD_PROBE_SAME_STATE=1 ABD_PROBE_THETA_ROW=5 python3 tools/probe_nuts_rate.py c3 4 200 > "$OUT/nuts_rate_c3_without_trains.txt" 2>&1
  # the kernel a NUTS chain's leapfrogs run (one chain per train unit): counters in their own pass
  ABD_PROBE_SAME_STATE=1 ABD_PROBE_THETA_ROW=5 ABD_PROBE_TUNE=30 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d "$OUT/train_pmc" --output-format csv -- python3 tools/probe_nuts_rate.py c3 4 10 > "$OUT/train_pmc.log" 2>&1
  python3 tools/pmc_summary.py "$OUT/train_pmc" abd_train > "$OUT/train_kernel_pmc_sq.txt" 2>&1
  ABD_PROBE_SAME_STATE=1 ABD_PROBE_THETA_ROW=5 ABD_PROBE_TUNE=30 rocprofv3 --pmc FETCH_SIZE -d "$OUT/train_pmc" --output-format csv -- python3 tools/probe_nuts_rate.py c3 4 10 > "$OUT/train_pmc.log" 2>&1
  python3 tools/pmc_summary.py "$OUT/train_pmc" abd_train > "$OUT/train_kernel_pmc_fetch_size.txt" 2>&1
  ABD_PROBE_SAME_STATE=1 ABD_PROBE_THETA_ROW=5 ABD_PROBE_TUNE=60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_trace" -- python3 tools/probe_nuts_rate.py c3 4 60 > "$OUT/train_trace.log" 2>&1
  find "$OUT/train_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/nuts_c3_kernel_stats.csv" \;
  rm -rf "$OUT/train_pmc" "$OUT/train_trace"
  # abdpymc-infer at BASELINE config 3's size: a cohort directory in the reference's format, thinned recording
  python3 tools/make_cohort_dir.py /tmp/abd_c3_cohort 10000 200 > "$OUT/cli_c3.txt" 2>&1
  python3 tools/run_with_rss.py python3 -m abdpymc_amd.cli --ititers_data /tmp/abd_c3_cohort --tune 100 --draws 200 --chains 4 --thin 50 --netcdf /tmp/abd_c3_post >> "$OUT/cli_c3.txt" 2>&1
  python3 -c "import numpy as np; z = np.load('/tmp/abd_c3_post.npz'); print({k: z[k].shape for k in ('p', 'i', 'ab_n_mu', 'mean_i', 'mean_ab_n_mu', 'mean_ab_s_mu', 'draw_index')})" >> "$OUT/cli_c3.txt" 2>&1
  rm -rf /tmp/abd_c3_cohort /tmp/abd_c3_post.npz
  python3 tools/probe_sync_latency.py > "$OUT/sync_latency_c3.txt" 2>&1
  for c in 4 16; do python3 tools/probe_nuts_rate.py default $c 300; done > "$OUT/nuts_rate_default_cohort.txt" 2>&1
  ABD_SAMPLER_UNIT=1 python3 tools/probe_dense_determinism.py 6 4 10000 200 > "$OUT/determinism_c3.txt" 2>&1
  ABD_SAMPLER_UNIT=2 python3 tools/probe_dense_determinism.py 6 8 10000 200 >> "$OUT/determinism_c3.txt" 2>&1
  python3 tools/bench_sampler.py default --chains 4 --tune 1000 --draws 1000 > "$OUT/sampler_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py default --chains 4 --tune 1000 --draws 1000 --dense >> "$OUT/sampler_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py default --chains 16 --tune 1000 --draws 1000 --no-record >> "$OUT/sampler_default_cohort.txt" 2>&1
  python3 tools/bench_sampler.py c3 --chains 4 --tune 100 --draws 100 --no-record > "$OUT/sampler_c3.txt" 2>&1
fi
python3 -c "import bench; print(bench.kernel_sources_sha256())" > "$OUT/kernel_sources_sha256.txt"
ls "$OUT" | wc -l
