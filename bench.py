#!/usr/bin/env python3
"""
bench.py -- logp+grad evaluations per second of the abdpymc joint log-probability on MI355X.

Metric (BASELINE.json): "logp+grad evals/sec, 10k-ind x 200-gap cohort".  Workload at N=1 is BASELINE
config 3: synthetic 10 000 individuals x 200 gaps, fp64, 4 chains on one GPU.  One *step* = one batched
logp+dlogp call for the rank's 4 chains at 4 fresh thetas (= 4 evaluations; the shared OD panels are
read once per launch).  For N>1 GPUs every rank holds its own 4 chains (config 4 at N=8: 32 chains,
weak scaling, no data-path collective) and the (steps x chains x 18) sample block is gathered over RCCL
once at the end of the timed region.

Timed region: K steps enqueued stream-ordered (the library rotates them over three HIP streams so that
launches share the chip instead of draining it one after the other), one wait, results fetched (host-side prior terms included),
bracketed by barrier + device sync.  Inputs are resident in HBM before it starts.  Warm-up: W untimed steps
as asked, repeated until at least 60 ms have passed -- after an idle period the part needs ~3 ms of load to
reach its sustained state, which 20 steps (0.6 ms) do not cover.

Also reported in the same JSON line:
  roofline     algorithmic bytes per launch / mean kernel time against 8 TB/s HBM.  The kernel time is taken with
               HIP events on the launch stream in an instrumented second pass over the same steps, in which
               launches are serialised on ONE stream (a launch's own duration means nothing while another
               one is in flight); `overlapped_us_per_launch` is the timed region's wall time per launch
  cpu_baseline the plain-C OpenMP restatement (oracle/abd_oracle.c) on the host cores, bounded sample
  sync_evals_per_s   rate seen by a caller that waits for every step (a sequential NUTS leapfrog chain)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured: 6.3 TB/s read stream, 4.7-5.2 TB/s copy (tools/micro/hbm_stream.hip)

CONFIGS = {
    "c2": dict(n_inds=1000, n_gaps=60, storage="f64", chains=4, name="synthetic 1000 ind x 60 gaps, fp64, 4 chains/GPU"),
    "c3": dict(n_inds=10000, n_gaps=200, storage="f64", chains=4, name="synthetic 10000 ind x 200 gaps, fp64, 4 chains/GPU"),
    "c5": dict(n_inds=100000, n_gaps=200, storage="f32", chains=1, name="synthetic 100000 ind x 200 gaps, fp32 storage, 1 chain/GPU"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the config's)")
    ap.add_argument("--splits", default="", help="comma separated gap indexes, e.g. 100 or 66,133")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.chains:
        cfg["chains"] = args.chains
    n_gpus = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus > 1 and world != n_gpus:
        raise SystemExit(
            f"--gpus {n_gpus} needs one process per GPU: python -m torch.distributed.run --nnodes=1 "
            f"--nproc-per-node {n_gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {n_gpus} ..."
        )

    dist = None
    torch = None
    if world > 1 or os.environ.get("ABD_BENCH_FORCE_DIST"):
        # torch is plumbing only: rendezvous, barrier and the RCCL gather of the sample block
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from abdpymc_amd import synthetic
    from abdpymc_amd._native import Context

    G, N, C = cfg["n_gaps"], cfg["n_inds"], cfg["chains"]
    splits = tuple(int(s) for s in args.splits.split(",") if s) or None
    K, W = args.steps, args.warmup

    sc = synthetic.make_cohort(N, G)
    ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=splits, n_chains=C,
                  storage=cfg["storage"], device=local_rank)
    chains = np.arange(C, dtype=np.int32)
    thetas = np.empty((K + W, C, 17))
    states = []
    for c in range(C):
        gchain = rank * C + c  # global chain id
        i_raw, w = synthetic.make_chain_state(N, G, gchain)
        ctx.set_discrete(c, i_raw, w)
        states.append((i_raw, w))
        thetas[:, c, :] = synthetic.make_thetas(G, K + W, gchain)
    nslots = ctx.n_result_slots

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.wait()

    def run_steps(lo, hi, out_lp=None, out_g=None):
        """enqueue steps [lo, hi) stream-ordered; fetch in windows of the result ring"""
        s = lo
        while s < hi:
            e = min(hi, s + nslots)
            for k in range(s, e):
                ctx.enqueue(k - s, chains, thetas[k])
            ctx.wait()
            lp, g = ctx.fetch_many(np.arange(e - s), C)  # host side: prior terms + scaling of the device sums
            if out_lp is not None:
                out_lp[s - lo:e - lo] = lp
                out_g[s - lo:e - lo] = g
            s = e

    # ---- warm-up: W steps, repeated until the part is at its sustained state ----
    run_steps(0, W)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.06:
        run_steps(0, max(W, 8) if W + K >= 8 else W + K)
    # ---- timed region: EXACTLY K steps ----
    lp_all = np.empty((K, C))
    g_all = np.empty((K, C, 17))
    if dist is not None:
        # RCCL builds its communicator and buffers on first use: not part of a step
        from abdpymc_amd.distributed import gather_samples

        gather_samples(np.zeros((K, C, 18)), dist, device="cuda")
    # The first barrier absorbs the collective's start-up; the part idles meanwhile and a few ms of idling cost
    # ~2.5 ms of ramp-up afterwards (tools/probe_idle_penalty.py), so: barrier, warm up again, then the barrier
    # that opens the timed region (a fraction of a millisecond by now).
    barrier()
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.015:
        run_steps(0, max(W, 8) if W + K >= 8 else W + K)
    t_b = time.perf_counter()
    barrier()
    if os.environ.get("ABD_BENCH_DEBUG"):
        print(f"[bench debug] opening barrier {(time.perf_counter() - t_b) * 1e3:.3f} ms", file=sys.stderr)
    t0 = time.perf_counter()
    run_steps(W, W + K, lp_all, g_all)
    if dist is not None:
        # the trivial RCCL gather of samples over xGMI: (world, steps, chains, 18) on every rank
        from abdpymc_amd.distributed import gather_samples

        gathered = gather_samples(np.concatenate([lp_all[..., None], g_all], axis=-1), dist, device="cuda")
        assert gathered.shape == (world, K, C, 18)
    t_steps = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("ABD_BENCH_DEBUG"):
        print(f"[bench debug] steps+gather {t_steps * 1e3:.3f} ms, closing barrier {(elapsed - t_steps) * 1e3:.3f} ms", file=sys.stderr)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if not np.all(np.isfinite(lp_all)):
        raise SystemExit("non-finite logp in the timed region")

    total_evals = K * C * world
    value = total_evals / elapsed

    # ---- synchronous caller rate (each step waits for its result) ----
    ks = min(K, 200)
    ctx.wait()
    t1 = time.perf_counter()
    for k in range(W, W + ks):
        ctx.logp_dlogp_batch(chains, thetas[k])
    sync_rate = ks * C / (time.perf_counter() - t1)

    # ---- kernel-only time: HIP events around every evaluation-kernel launch, same steps ----
    ctx.kernel_timing(True)
    ctx.kernel_time(reset=True)
    run_steps(W, W + K)
    k_ms, k_n = ctx.kernel_time(reset=True)
    ctx.kernel_timing(False)
    alg_bytes = ctx.algorithmic_bytes(C)  # compulsory bytes of this library's layout (bit-packed indicators)
    R = 4 if cfg["storage"] == "f32" else 8
    survey_bytes = G * N * (4 * R + 2 + C) + C * N  # byte-per-cell indicator panels
    k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
    achieved = alg_bytes / k_avg_s / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get(args.config, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = dict(
        bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
        traffic=traffic, kernel="abd_dense_kernel", kernel_us=round(k_avg_s * 1e6, 3), launches=int(k_n),
        algorithmic_bytes_per_launch=int(alg_bytes), survey_bytes_per_launch=int(survey_bytes), evals_per_launch=C,
        overlapped_us_per_launch=round(elapsed / K * 1e6, 3),
        note="achieved uses the smaller, bit-packed byte count; the kernel is fp64-VALU bound (see DESIGN.md). kernel_us is "
             "ONE launch alone on the chip (instrumented pass, one stream, full grid); in the timed region three launches "
             "share the chip (3 streams x 1 workgroup per CU) and one completes every overlapped_us_per_launch",
    )

    # ---- CPU baseline: plain-C OpenMP restatement on the host cores (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import abd_oracle as O
        from oracle import c_oracle

        coh = O.Cohort(G, N, sc.vacs, sc.pcrpos, O.AntigenObs(sc.idx_gap, sc.idx_ind, sc.x_s, sc.y_s),
                       O.AntigenObs(sc.idx_gap, sc.idx_ind, sc.x_n, sc.y_n))
        co = c_oracle.COracle(coh, splits)
        # the GPU box gives one GPU's share of the host (16 cores); more OpenMP threads than that only thrash
        cores = max(1, min(c_oracle.max_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("ABD_CPU_THREADS", "16"))))
        i_raw, w = states[0]
        lp_c, g_c = co.logp_dlogp(thetas[W, 0], i_raw, w, nthreads=cores)  # warm + parity spot check of the bench itself
        scale = np.maximum(np.abs(g_c), 1e-6 * np.abs(g_c).max())
        if abs(lp_c - lp_all[0, 0]) > 1e-6 * abs(lp_c) or (np.abs(g_all[0, 0] - g_c) / scale).max() > 1e-6:
            raise SystemExit(f"bench parity check failed: gpu {lp_all[0, 0]} vs cpu {lp_c}")
        n_done, t2 = 0, time.perf_counter()
        while True:
            co.logp_dlogp(thetas[W + (n_done % K), 0], i_raw, w, nthreads=cores)
            n_done += 1
            el = time.perf_counter() - t2
            if el >= args.cpu_seconds or n_done >= 20000:
                break
        cpu = dict(value=round(n_done / el, 3), unit="evals/s", cores=cores, kind="port",
                   sample=f"{n_done} logp+grad evals of chain 0 at fresh thetas on the same cohort ({el:.1f} s), "
                          f"oracle/abd_oracle.c, OpenMP {cores} threads")

    # ---- the whole compound step the path serves (informative; N=1 only): NUTS + Gibbs sweep, all chains in lock step
    compound = None
    if rank == 0 and world == 1:
        iters = 12
        smp = ctx.sampler(chains, thetas[W], tune=iters, seed=1)
        t3 = time.perf_counter()
        _, st = smp.run(iters)
        dt3 = time.perf_counter() - t3
        t4 = time.perf_counter()
        ctx.gibbs_sweep(chains, thetas[W], seed=1, sweep=0)
        sweep_ms = (time.perf_counter() - t4) * 1e3
        smp.close()
        compound = dict(chain_iterations_per_s=round(iters * C / dt3, 1), iterations=iters,
                        leapfrogs_per_iteration=round(float(st["n_steps"].mean()), 1), gibbs_sweep_ms=round(sweep_ms, 3),
                        note="abd_sampler_run from the bench's chain states, early tuning (step size still adapting)")

    if rank == 0:
        line = {
            "metric": "logp+grad evals/sec",
            "value": round(value, 1),
            "unit": "evals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "n_inds": N, "n_gaps": G, "storage": cfg["storage"],
                       "chains_per_gpu": C, "total_chains": C * world, "splits": list(splits or ()),
                       "evals_per_step_per_gpu": C, "parallelism": f"chains sharded {C}/GPU x {world}"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "sync_evals_per_s": round(sync_rate, 1),
            "compound_step": compound,
            "device": ctx.device_name,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
