#!/usr/bin/env python3
"""
bench.py -- logp+grad evaluations per second of the abdpymc joint log-probability on MI355X.

Metric (BASELINE.json): "logp+grad evals/sec, 10k-ind x 200-gap cohort".  Workload at N=1 is BASELINE
config 3: synthetic 10 000 individuals x 200 gaps, fp64, 4 chains on one GPU.  One *step* = one batched
logp+dlogp call for the rank's 4 chains at 4 fresh thetas (= 4 evaluations; the shared OD panels are
read once per launch).  For N>1 GPUs every rank holds its own 4 chains (config 4 at N=8: 32 chains,
weak scaling, no data-path collective); the (steps x chains x 18) sample block is gathered over RCCL
once after the timed region and timed on its own (`gather_ms`).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks
(`python -m torch.distributed.run ...`, before this process has touched the GPU) and passes rank 0's line on.

Timed region: K steps through `abd_logp_dlogp_many` -- enqueued stream-ordered (the library rotates them over four HIP
streams so that launches share the chip instead of draining it one after the other), one wait, results fetched
(host-side prior terms included) -- opened by barrier + device sync, closed by a device sync; MAX over ranks.  Inputs
are resident in HBM before it starts.  The K-step region is repeated R times (until >= 0.25 s of timed work,
R <= 400): `value` and `ms_per_step` are the MEDIAN region, `region_ms` holds min / median / max.  Warm-up:
W untimed steps as asked, repeated until at least 60 ms have passed -- after an idle period the part needs
~3 ms of load to reach its sustained state, which 20 steps (0.5 ms) do not cover.

Also in the same JSON line:
  roofline     the TIMED launch shape: device time per launch from HIP events around every window of K
               stream-ordered launches (first launch .. all streams joined; abd_kernel_timing mode 2), against
               8 TB/s HBM with the algorithmic bytes of one launch.  `roofline.isolated` is ONE launch alone on
               the chip (one stream, full grid; mode 1), `roofline.valu` the roof that actually binds: vector
               instructions per launch (rocprofv3 PMC, profiles/) against the fp64 vector peak
  cpu_baseline the plain-C OpenMP restatement (oracle/abd_oracle.c) on the host cores, its 1-thread rate, and B0,
               the reference's own dense (G, G, N) algorithm restated in NumPy (forward only), bounded samples
  sync_evals_per_s   rate seen by a caller that waits for every step (lock-step NUTS over the rank's chains)
  nuts_evals_per_s   the DELIVERED rate: leapfrogs per second of the native sampler's NUTS over the config's chains
                     (abd_sampler_run without the sweep, step size settled, >= 1 s or >= 200 iterations): a chain's
                     leapfrogs follow each other (abd.py:922), so this -- not `value` -- is what a sampling run gets
  wait_fallbacks     completion-tag waits that fell back to a stream synchronise during the whole run (expect 0)

--config c1 is BASELINE config 1, the reference's default cohort (1 520 individuals x 31 gaps, 35 709 OD readings kept as
observation lists; tests/golden/default_cohort.npz), 4 chains per call: ~0.9 MB per launch, bound by launches, not bytes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured: 6.3 TB/s read stream (tools/micro/hbm_stream.hip)
FP64_VECTOR_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes/clk x 2 flop x 2.4 GHz (MI355X_MICROARCH.md: half the fp32 vector rate)

CONFIGS = {
    "c1": dict(n_inds=1520, n_gaps=31, storage="f64", chains=4, cohort="default",
               name="reference default cohort: 1520 ind x 31 gaps, 35709 OD readings (observation lists), fp64, 4 chains/GPU"),
    "c2": dict(n_inds=1000, n_gaps=60, storage="f64", chains=4, name="synthetic 1000 ind x 60 gaps, fp64, 4 chains/GPU"),
    "c3": dict(n_inds=10000, n_gaps=200, storage="f64", chains=4, name="synthetic 10000 ind x 200 gaps, fp64, 4 chains/GPU"),
    "c5": dict(n_inds=100000, n_gaps=200, storage="f32", chains=1, name="synthetic 100000 ind x 200 gaps, fp32 storage, 1 chain/GPU"),
}


def kernel_sources_sha256():
    """sha256 over the device sources the PMC counters of profiles/traffic.json were collected on"""
    import hashlib

    h = hashlib.sha256()
    for f in ("abd_types.hpp", "abd_device.hpp", "abd_dense.hpp", "abd_obs.hpp", "abd_sparse.hpp"):
        h.update(open(os.path.join(ROOT, "abdpymc_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_gpus):
    """Start one rank per GPU under torch.distributed.run.  Nothing in this process has touched HIP yet."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def build_context(cfg, device, rank=0, splits=None):
    """(ctx, cohort, splits, states, G, N, C) of a BASELINE configuration on `device`: panels resident, every chain slot holding its
    own random discrete state."""
    from types import SimpleNamespace

    from abdpymc_amd import synthetic
    from abdpymc_amd._native import Context

    G, N, C = cfg["n_gaps"], cfg["n_inds"], cfg["chains"]
    if cfg.get("cohort") == "default":
        # BASELINE config 1: the reference's own cohort (data/cohort_data, packed as tests/golden/default_cohort.npz) with the
        # splits abdpymc-infer --split_delta --split_omicron gives it (abd.py:204-221)
        z = np.load(os.path.join(ROOT, "tests", "golden", "default_cohort.npz"))
        m_s = z["is_s"]
        cols = lambda m: (z["elapsed_months"][m], z["individual_i"][m], z["log_dilution"][m], z["od"][m])  # noqa: E731
        sc = SimpleNamespace(s_obs=cols(m_s), n_obs=cols(~m_s), vacs=z["vacs"], pcrpos=z["pcrpos"])
        assert int(z["elapsed_months"].max()) + 1 == G and int(z["individual_i"].max()) + 1 == N
        splits = splits or (14, 20)
    else:
        sc = synthetic.make_cohort(N, G)
    ctx = Context(G, N, sc.s_obs, sc.n_obs, sc.vacs, sc.pcrpos, splits=splits, n_chains=C, storage=cfg["storage"], device=device)
    states = []
    for c in range(C):
        i_raw, w = synthetic.make_chain_state(N, G, rank * C + c)  # (global chain id)
        ctx.set_discrete(c, i_raw, w)
        states.append((i_raw, w))
    return ctx, sc, splits, states, G, N, C


def nuts_rates(ctx, chains, theta0, tune=100, iters=200, seed=3, pooled=False):
    """The rate a sampling run gets: leapfrogs (= logp+grad evaluations) per wall second inside abd_sampler_run, NUTS only.
    `tune` adapting iterations (untimed), then `iters` timed ones at the adapted step sizes.  Chains are independent and
    their trees differ, so they finish the call at different times: `value` is over the whole call (which ends with the
    slowest chain), `all_chains_at_work` over the stretch in which none has finished yet (from the per-iteration
    completion times the sampler reports).  pooled: after tuning every chain gets the SAME step size and metric (the
    geometric mean over the chains), so that the chains' trees are equally long on average."""
    C = len(chains)
    smp = ctx.sampler(chains, theta0, tune=tune, seed=seed, gibbs=False)
    smp.run(tune)
    if pooled:
        ad = [smp.adaptation(k) for k in range(C)]
        im = np.exp(np.mean([np.log(a[0]) for a in ad], axis=0))
        eps = float(np.exp(np.mean([np.log(a[1]) for a in ad])))
        for k in range(C):
            smp.set_adaptation(k, im, eps)
    t5 = time.perf_counter()
    _, st = smp.run(iters)
    t_n = time.perf_counter() - t5
    smp.close()
    n_lf = float(st["n_steps"].sum())
    t_first = float(st["t_done"][:, -1].min())
    in_window = float(st["n_steps"][st["t_done"] <= t_first].sum())
    return dict(value=round(n_lf / t_n, 1), all_chains_at_work=round(in_window / max(t_first, 1e-9), 1),
                first_chain_done_at=round(t_first / t_n, 3), chains=C, iterations=iters, tune=tune, seconds=round(t_n, 3),
                leapfrogs_per_iteration_and_chain=round(n_lf / iters / C, 1),
                leapfrogs_per_chain=[int(x) for x in np.asarray(st["n_steps"]).sum(axis=1)],
                us_per_leapfrog_of_a_chain=round(t_n / max(n_lf / C, 1.0) * 1e6, 2))


def other_config(key, device, K=20, W=5, min_seconds=0.15):
    """A bounded pass over another single-GPU BASELINE configuration: the stream-ordered rate of K-step regions (median),
    the device time per launch of that shape (HIP events, abd_kernel_timing mode 2) against the HBM roof, the waiting caller's
    rate, and the rate NUTS sees."""
    from abdpymc_amd import synthetic

    t_cfg = time.perf_counter()
    cfg = dict(CONFIGS[key])
    ctx, sc, splits, states, G, N, C = build_context(cfg, device)
    chains = np.arange(C, dtype=np.int32)
    thetas = np.empty((K + W, C, 17))
    for c in range(C):
        thetas[:, c, :] = synthetic.make_thetas(G, K + W, c)
    lp, g = np.empty((K, C)), np.empty((K, C, 17))
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        ctx.logp_dlogp_many(chains, thetas[:W], lp[:W], g[:W])
    ctx.kernel_timing(2)
    ctx.kernel_time(reset=True)
    times, per_launch = [], []
    t_r = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_r < min_seconds and len(times) < 200):
        ctx.wait()
        ctx.kernel_time(reset=True)
        t0 = time.perf_counter()
        ctx.logp_dlogp_many(chains, thetas[W:], lp, g)
        times.append(time.perf_counter() - t0)
        ms_r, n_r = ctx.kernel_time(reset=True)
        per_launch.append(ms_r / max(n_r, 1))
    ctx.kernel_timing(0)
    if not np.all(np.isfinite(lp)):
        raise SystemExit(f"non-finite logp in config {key}")
    el = float(np.median(times))
    k_us = float(np.median(per_launch)) * 1e3
    alg = ctx.algorithmic_bytes(C)
    ach = alg / (k_us * 1e-6) / 1e9
    ks = min(K, 20)
    t1 = time.perf_counter()
    for k in range(W, W + ks):
        ctx.logp_dlogp_batch(chains, thetas[k])
    sync = ks * C / (time.perf_counter() - t1)
    nuts = nuts_rates(ctx, chains, thetas[W], tune=60, iters=60)
    out = dict(workload=cfg["name"], value=round(K * C / el, 1), ms_per_step=round(el / K * 1e3, 5), steps=K, repeats=len(times),
               kernel_us=round(k_us, 3), roofline=dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, frac=round(ach / HBM_PEAK_GBS, 4),
                                                       algorithmic_bytes_per_launch=int(alg), evals_per_launch=C,
                                                       kernel="abd_dense_kernel" if ctx.is_dense else "abd_obs_kernel"),
               sync_evals_per_s=round(sync, 1), nuts_evals_per_s=nuts["value"], nuts=nuts, wait_fallbacks=int(ctx.wait_fallbacks))
    ctx.close()
    out["seconds"] = round(time.perf_counter() - t_cfg, 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the config's)")
    ap.add_argument("--splits", default="", help="comma separated gap indexes, e.g. 100 or 66,133")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="skip the sweep / compound-step / NUTS section (A/B runs of the kernels: tools/ab_bench.sh)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the bounded passes over BASELINE configs 5, 1 and 2 (`other_configs`)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--min-seconds", type=float, default=0.25, help="repeat the K-step region until this much timed work")
    ap.add_argument("--max-repeats", type=int, default=400)
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.chains:
        cfg["chains"] = args.chains
    n_gpus = args.gpus
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(n_gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: start one process per GPU (or let bench.py start them: "
                         f"python bench.py --gpus {n_gpus})")

    dist = None
    torch = None
    backend = None
    device = local_rank
    if world > 1 or os.environ.get("ABD_BENCH_FORCE_DIST"):
        # torch is plumbing only: rendezvous, barrier and the gather of the sample block
        import torch
        import torch.distributed as dist

        backend = os.environ.get("ABD_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; "gloo": rehearsal with ranks sharing a GPU
        n_dev = max(1, torch.cuda.device_count())
        device = local_rank % n_dev
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
    comm_dev = "cuda" if backend == "nccl" else None

    from abdpymc_amd import synthetic
    from abdpymc_amd._native import Context

    G, N, C = cfg["n_gaps"], cfg["n_inds"], cfg["chains"]
    splits = tuple(int(s) for s in args.splits.split(",") if s) or None
    K, W = args.steps, args.warmup

    default_cohort = cfg.get("cohort") == "default"
    ctx, sc, splits, states, G, N, C = build_context(cfg, device, rank, splits)
    chains = np.arange(C, dtype=np.int32)
    thetas = np.empty((K + W, C, 17))
    for c in range(C):
        thetas[:, c, :] = synthetic.make_thetas(G, K + W, rank * C + c)

    def barrier():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()
        ctx.wait()

    def all_max(values):
        """element-wise MAX over ranks of a vector of per-rank times"""
        if dist is None:
            return np.asarray(values, dtype=np.float64)
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        if comm_dev:
            t = t.to(comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy()

    scratch_lp = np.empty((K + W, C))
    scratch_g = np.empty((K + W, C, 17))

    def run_steps(lo, hi, out_lp=None, out_g=None):
        """steps [lo, hi) through the library's batched stream-ordered entry point (abd_logp_dlogp_many): every step is
        enqueued, the library waits once and fetches (host side: prior terms + scaling of the device sums) straight
        into the caller's arrays"""
        if hi <= lo:
            return
        if out_lp is None:
            out_lp, out_g = scratch_lp[:hi - lo], scratch_g[:hi - lo]
        ctx.logp_dlogp_many(chains, thetas[lo:hi], out_lp, out_g)

    def warm(seconds):
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < seconds:
            run_steps(0, max(W, 8) if W + K >= 8 else W + K)

    # ---- warm-up: W steps, repeated until the part is at its sustained state ----
    run_steps(0, W)
    warm(0.06)
    lp_all = np.empty((K, C))
    g_all = np.empty((K, C, 17))
    if dist is not None:
        # the communicator and its buffers are built on first use: not part of a step
        from abdpymc_amd.distributed import gather_samples

        gather_samples(np.zeros((K, C, 18)), dist, device=comm_dev)

    # ---- timed: EXACTLY K steps per region, R regions ----
    # The device time of every timed region is taken in the SAME region: HIP events on the streams the kernels run on
    # (abd_kernel_timing mode 2: a start event in front of the region's first launch, an end event per stream behind its
    # last launch and that launch's sum).  They are read back after the region's clock has stopped.
    ctx.kernel_timing(2)
    ctx.kernel_time(reset=True)
    w_per_launch, w_n = [], 0

    def region():
        # barrier, W steps so the part is under load again (a barrier idles it; a few ms of idling cost ~2.5 ms of
        # ramp-up, tools/probe_idle_penalty.py), then the barrier + sync that opens the region
        nonlocal w_n
        if dist is not None:
            barrier()
            run_steps(0, W)
        barrier()
        ctx.kernel_time(reset=True)
        t0 = time.perf_counter()
        run_steps(W, W + K, lp_all, g_all)  # every step's result waited for and fetched
        dt = time.perf_counter() - t0
        ms_r, n_r = ctx.kernel_time(reset=True)
        w_per_launch.append(ms_r / max(n_r, 1))
        w_n += n_r
        return dt

    def all_min(values):
        if dist is None:
            return np.asarray(values, dtype=np.float64)
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        if comm_dev:
            t = t.to(comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.cpu().numpy()

    first = float(all_max([region()])[0])
    n_rep = int(min(args.max_repeats, max(3, np.ceil(args.min_seconds / max(first, 1e-9)))))
    w_per_launch, w_n = [], 0
    own_times = np.array([region() for _ in range(n_rep)])
    times = all_max(own_times)
    times_fastest_rank = all_min(own_times)  # a straggler shows as a gap between the two
    elapsed = float(np.median(times))
    ctx.kernel_timing(0)
    if not np.all(np.isfinite(lp_all)):
        raise SystemExit("non-finite logp in the timed region")

    gather_ms = None
    if dist is not None:
        # the trivial gather of samples (RCCL over xGMI with backend nccl): (world, steps, chains, 18) on every rank
        barrier()
        t_g = time.perf_counter()
        gathered = gather_samples(np.concatenate([lp_all[..., None], g_all], axis=-1), dist, device=comm_dev)
        gather_ms = float(all_max([time.perf_counter() - t_g])[0]) * 1e3
        assert gathered.shape == (world, K, C, 18)

    total_evals = K * C * world
    value = total_evals / elapsed

    # ---- synchronous caller rate (each step waits for its result): what lock-step NUTS over these chains sees ----
    ks = min(K, 200)
    sync_rates = []
    for _ in range(3):
        ctx.wait()
        t1 = time.perf_counter()
        for k in range(W, W + ks):
            ctx.logp_dlogp_batch(chains, thetas[k])
        sync_rates.append(ks * C / (time.perf_counter() - t1))
    sync_rate = float(np.median(sync_rates))

    # ---- the isolated kernel: HIP events around every launch, one stream, full grid ----
    warm(0.02)
    ctx.kernel_timing(1)
    ctx.kernel_time(reset=True)
    k_per_launch, k_n = [], 0
    for _ in range(int(min(n_rep, 10))):
        run_steps(W, W + K)
        ms_r, n_r = ctx.kernel_time(reset=True)
        k_per_launch.append(ms_r / max(n_r, 1))
        k_n += n_r
    ctx.kernel_timing(0)
    # medians over the repetitions, like the timed regions (a stray slow window would otherwise carry the mean)
    w_avg_s = float(np.median(w_per_launch)) * 1e-3
    k_avg_s = float(np.median(k_per_launch)) * 1e-3

    alg_bytes = ctx.algorithmic_bytes(C)  # compulsory bytes of this library's layout (bit-packed indicators)
    R = 4 if cfg["storage"] == "f32" else 8
    survey_bytes = G * N * (4 * R + 2 + C) + C * N  # byte-per-cell indicator panels
    achieved = alg_bytes / w_avg_s / 1e9
    iso_achieved = alg_bytes / k_avg_s / 1e9
    prof, prof_stale = {}, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp) and (not splits or default_cohort) and C == CONFIGS[args.config]["chains"]:
        try:
            whole = json.load(open(tp))
            prof = whole.get(args.config, {})
            # the counters were collected on a build of these sources: anything else and they describe another kernel
            prof_stale = whole.get("kernel_sources_sha256") != kernel_sources_sha256()
        except Exception:
            prof = {}
    pipe_prof = prof.get("pipe_grid", prof)
    valu = None
    if pipe_prof.get("valu_insts_per_launch"):
        insts = float(pipe_prof["valu_insts_per_launch"])
        tf = insts * 64 * 2 / w_avg_s / 1e12  # every vector instruction priced as one fp64 FMA on 64 lanes
        valu = dict(bound="fp64-valu", stale=bool(prof_stale), insts_per_launch=int(insts), achieved_tflops_equiv=round(tf, 2), peak=FP64_VECTOR_TFLOPS,
                    unit="TFLOP/s", frac=round(tf / FP64_VECTOR_TFLOPS, 4),
                    insts_per_cell_chain=round(insts * 64 / (G * N * C), 2), source=pipe_prof.get("valu_source"),
                    note="vector wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, profiles/) x 64 lanes x 2 flop / device "
                         "time per launch of the timed shape; 32-bit integer/select instructions issue at twice the fp64 rate, "
                         "so 1.0 is not reachable and the lever is the instruction count")
    roofline = dict(
        bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
        traffic=pipe_prof.get("hbm_bytes_per_launch"), traffic_stale=bool(prof_stale) if pipe_prof.get("hbm_bytes_per_launch") else None,
        kernel="abd_dense_kernel" if ctx.is_dense else "abd_obs_kernel",
        kernel_us=round(w_avg_s * 1e6, 3), launches=int(w_n),
        launch_shape=f"stream-ordered, as timed: launches rotate over {ctx.n_pipes} HIP streams on different hardware queues, "
                     "1 workgroup per CU each; device time from HIP events around the K launches of every timed region / K (median)",
        algorithmic_bytes_per_launch=int(alg_bytes), survey_bytes_per_launch=int(survey_bytes), evals_per_launch=C,
        isolated=dict(kernel_us=round(k_avg_s * 1e6, 3), achieved=round(iso_achieved, 2), frac=round(iso_achieved / HBM_PEAK_GBS, 4),
                      launches=int(k_n), traffic=prof.get("hbm_bytes_per_launch"),
                      launch_shape="one launch alone on the chip: one stream, full grid (4 workgroups per CU); what a "
                                   "synchronous call runs"),
        valu=valu,
        note=("achieved = algorithmic_bytes_per_launch / kernel_us (the smaller, bit-packed byte count). The working set "
              "(66 MB at config 3) sits in the 256 MiB Infinity Cache and the kernel is fp64-VALU bound: see roofline.valu")
        if ctx.is_dense else "observation lists: ~0.9 MB per launch, bound by the two launches of an evaluation, not by bytes",
    )

    # ---- CPU baselines on the host cores (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import abd_oracle as O
        from oracle import c_oracle

        def cohort_of(s):
            return O.Cohort(s.n_gaps, s.n_inds, s.vacs, s.pcrpos, O.AntigenObs(s.idx_gap, s.idx_ind, s.x_s, s.y_s),
                            O.AntigenObs(s.idx_gap, s.idx_ind, s.x_n, s.y_n))

        def rate(fn, budget, max_n=20000):
            n_done, t2 = 0, time.perf_counter()
            while True:
                fn(n_done)
                n_done += 1
                el = time.perf_counter() - t2
                if el >= budget or n_done >= max_n:
                    return n_done / el, n_done, el

        if default_cohort:
            coh = O.Cohort(G, N, sc.vacs, sc.pcrpos, O.AntigenObs(*sc.s_obs), O.AntigenObs(*sc.n_obs))
        else:
            coh = cohort_of(sc)
        co = c_oracle.COracle(coh, splits)
        # the GPU box gives one GPU's share of the host (16 cores); more OpenMP threads than that only thrash
        cores = max(1, min(c_oracle.max_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("ABD_CPU_THREADS", "16"))))
        i_raw, w = states[0]
        lp_c, g_c = co.logp_dlogp(thetas[W, 0], i_raw, w, nthreads=cores)  # warm + parity spot check of the bench itself
        scale = np.maximum(np.abs(g_c), 1e-6 * np.abs(g_c).max())
        if abs(lp_c - lp_all[0, 0]) > 1e-6 * abs(lp_c) or (np.abs(g_all[0, 0] - g_c) / scale).max() > 1e-6:
            raise SystemExit(f"bench parity check failed: gpu {lp_all[0, 0]} vs cpu {lp_c}")
        budget = args.cpu_seconds
        v_n, n_n, el_n = rate(lambda k: co.logp_dlogp(thetas[W + (k % K), 0], i_raw, w, nthreads=cores), 0.5 * budget)
        v_1, n_1, el_1 = rate(lambda k: co.logp_dlogp(thetas[W + (k % K), 0], i_raw, w, nthreads=1), 0.25 * budget)
        # B0: the reference's algorithm as the reference states it -- dense (G, G, N) decay design, forward logp only
        # (the reference adds reverse-mode autodiff over the same graph) -- at config 2, where it takes ~35 ms
        sc2 = synthetic.make_cohort(CONFIGS["c2"]["n_inds"], CONFIGS["c2"]["n_gaps"])
        coh2 = cohort_of(sc2)
        i2, w2 = synthetic.make_chain_state(sc2.n_inds, sc2.n_gaps, 0)
        th2 = synthetic.make_thetas(sc2.n_gaps, 1, 0)[0]
        O.joint_logp(th2, i2, w2, coh2, dense=True)
        v_b0, n_b0, el_b0 = rate(lambda k: O.joint_logp(th2, i2, w2, coh2, dense=True), 0.2 * budget, 200)
        b0 = dict(value=round(v_b0, 3), unit="evals/s (forward logp only)", cores=1, config="c2: 1000 ind x 60 gaps",
                  sample=f"{n_b0} evaluations in {el_b0:.1f} s, oracle/abd_oracle.py joint_logp(dense=True): the (G, G, N) form of abd.py:242-274",
                  c3_recorded="4.0 s per forward evaluation at 10000 x 200 (3.2 GB of temporaries; tools/cpu_baselines.py on a GPU box, DESIGN.md section 5)")
        cpu = dict(value=round(v_n, 3), unit="evals/s", cores=cores, kind="port",
                   sample=f"{n_n} logp+grad evals of chain 0 at fresh thetas on the same cohort ({el_n:.1f} s), "
                          f"oracle/abd_oracle.c, OpenMP {cores} threads",
                   one_thread=dict(value=round(v_1, 3), unit="evals/s", cores=1, sample=f"{n_1} evals in {el_1:.1f} s, same code, 1 thread"),
                   b0_reference_algorithm=b0)

    # ---- the rate a sampling run gets, and the compound step around it (rank 0, N=1 only) ----
    compound, nuts, nuts_own, nuts_bal = None, None, None, None
    if rank == 0 and world == 1 and not args.no_sampler:
        def sweep_ms(theta_rows, n=5):
            ts = []
            for k in range(n):
                t4 = time.perf_counter()
                ctx.gibbs_sweep(chains, theta_rows, seed=1, sweep=k)
                ts.append((time.perf_counter() - t4) * 1e3)
            return float(np.median(ts))

        # one sweep of all chains from the bench's FRESH random discrete state (far from the posterior: long walks) --
        # measured before any sampler run has rewritten it; the first sweep of the five starts from exactly that state
        for c in range(C):
            ctx.set_discrete(c, *states[c])
        sweep_random = sweep_ms(thetas[W])
        sweep_conv = None
        if not default_cohort:
            # ... and on a converged chain: the simulation's own parameters and infections (almost every proposal is
            # rejected within a few gaps there)
            for c in range(C):
                ctx.set_discrete(c, sc.i_true, np.ones(N, dtype=np.int8))
            sweep_conv = sweep_ms(np.tile(synthetic.truth_theta(G), (C, 1)))
        for c in range(C):
            ctx.set_discrete(c, *states[c])
        # the compound step as abdpymc-infer runs it (NUTS transition, sweep, re-evaluation; nothing recorded): 200 iterations of
        # every chain from the bench's random chain states, the first 100 adapting
        n_cs = 200
        smp = ctx.sampler(chains, thetas[W], tune=100, seed=1)
        t3 = time.perf_counter()
        _, st = smp.run(n_cs)
        dt3 = time.perf_counter() - t3
        smp.close()
        compound = dict(chain_iterations_per_s=round(n_cs * C / dt3, 1), iterations=n_cs, seconds=round(dt3, 3),
                        leapfrogs_per_iteration=round(float(st["n_steps"].mean()), 1),
                        gibbs_acceptances_per_sweep=round(float(st["gibbs_accepted"].mean()), 1),
                        gibbs_sweep_ms=round(sweep_random, 3),
                        gibbs_sweep_ms_converged_state=None if sweep_conv is None else round(sweep_conv, 3),
                        note="chain_iterations_per_s: abd_sampler_run, NUTS + sweep + re-evaluation, 200 iterations of every chain from the "
                             "bench's fresh random chain states (100 adapting), nothing recorded; gibbs_sweep_ms: one sweep of all chains "
                             f"(median of 5) starting from that fresh random discrete state, before any sampler run ({C} chains x {G * N + N} "
                             "binary dims); _converged_state: the same on the simulation's own infections and parameters")
        # NUTS as the native sampler runs it (no sweep): 100 adapting iterations, then one call of 200 at the adapted step sizes.
        # `nuts`: every chain samples the SAME target (chain 0's discrete state; own start, own random stream), as the chains of
        # a real run do once they have converged; `nuts_own_states`: every chain on its own random discrete state -- without a
        # sweep those are four different targets whose step sizes, and with them the tree lengths, differ several-fold
        for c in range(C):
            ctx.set_discrete(c, *states[0])
        nuts = nuts_rates(ctx, chains, thetas[W])
        for c in range(C):
            ctx.set_discrete(c, *states[c])
        nuts_own = nuts_rates(ctx, chains, thetas[W])
        for c in range(C):
            ctx.set_discrete(c, *states[0])
        nuts_bal = nuts_rates(ctx, chains, thetas[W], pooled=True)
        for c in range(C):
            ctx.set_discrete(c, *states[c])
        note = ("leapfrogs (= logp+grad evaluations) of all chains per wall second inside abd_sampler_run, NUTS only; dense cohorts "
                "run leapfrog trains (abd_train.hpp: a launch takes the chains of its unit one leapfrog further and leaves the next "
                "points for the launch queued behind it).  Chains are independent: `value` is over the whole call, which ends with "
                "the chain that had the most leapfrogs (leapfrogs_per_chain), `all_chains_at_work` over the stretch before the "
                "first chain finishes")
        nuts["note"] = note + "; every chain on the same discrete state (chain 0's)"
        nuts_own["note"] = note + "; every chain on its own random discrete state"
        nuts_bal["note"] = note + ("; every chain on the same discrete state and, after tuning, on the same step size and metric (pooled "
                                   "over the chains: abd_sampler_set_adaptation), so that no chain idles behind another for long")

    wait_fallbacks, device_name = int(ctx.wait_fallbacks), ctx.device_name
    other = None
    if rank == 0 and world == 1 and args.config == "c3" and not args.no_other_configs and not args.no_sampler:
        ctx.close()
        ctx = None
        other = {key: other_config(key, device) for key in ("c5", "c1", "c2")}

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
            "dtype": "f64",  # the type the path computes in; with config.storage = "f32" only the panels in HBM are fp32
            "data": "synthetic",
            "config": {"workload": cfg["name"], "n_inds": N, "n_gaps": G, "storage": cfg["storage"],
                       "chains_per_gpu": C, "total_chains": C * world, "splits": list(splits or ()),
                       "evals_per_step_per_gpu": C, "parallelism": f"chains sharded {C}/GPU x {world}"},
            "value_kind": "stream-ordered: the K steps of a region are enqueued back to back at independent thetas and "
                          "waited for once; sync_evals_per_s is the rate when every step is waited for",
            "repeats": int(n_rep),
            "region_ms": {"min": round(float(times.min()) * 1e3, 4), "median": round(elapsed * 1e3, 4),
                          "max": round(float(times.max()) * 1e3, 4), "first": round(first * 1e3, 4),
                          "median_fastest_rank": round(float(np.median(times_fastest_rank)) * 1e3, 4)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "sync_evals_per_s": round(sync_rate, 1),
            "nuts_evals_per_s": None if nuts is None else nuts["value"],
            "nuts": nuts,
            "nuts_own_states": nuts_own,
            "nuts_balanced": nuts_bal,
            "compound_step": compound,
            "sampler_c3" if args.config == "c3" else "sampler": None if compound is None else
            {k: compound[k] for k in ("chain_iterations_per_s", "iterations", "seconds", "leapfrogs_per_iteration", "gibbs_acceptances_per_sweep")},
            "other_configs": other,
            "wait_fallbacks": wait_fallbacks,
            "gather_ms": None if gather_ms is None else round(gather_ms, 4),
            "dist": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                "device_of_rank0": device},
            "device": device_name,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
