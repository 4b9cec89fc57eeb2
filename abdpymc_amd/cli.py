"""
``abdpymc-infer``: same flags and flow as the reference entry point (abd.py:885-924) --
TiterData.from_disk -> calculate_splits -> model(...) -> sample(tune, draws) -> write the posterior.

With PyMC installed the model can instead be handed to ``pm.sample`` through
:mod:`abdpymc_amd.pytensor_op`; this command uses the built-in compound sampler (NUTS + binary Gibbs) so it
runs on a box that has neither PyMC nor ArviZ.  Output: ArviZ NetCDF when ArviZ is importable, else a
``.npz`` with the same variable names (leading axes chain, draw; Deterministics with trailing gap, ind).
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser("abdpymc-infer")
    # reference flags, verbatim (abd.py:888-911)
    parser.add_argument("--tune", help="Number of tuning steps.", type=int, required=True)
    parser.add_argument("--draws", help="Number of draws.", type=int, required=True)
    parser.add_argument("--cores", help="Number of cores", type=int)
    parser.add_argument("--ititers_data", help="Path to directory for generating TiterData object.", default="cohort_data")
    parser.add_argument("--split_delta", help="Split time chunk between delta and pre-delta", action="store_true")
    parser.add_argument("--split_omicron", help="Split time chunk between omicron and delta", action="store_true")
    parser.add_argument("--ignore_pcrpos", help="Ignore PCR+ data", action="store_true")
    parser.add_argument("--netcdf", help="Path of netCDF file to save.")
    # additions (all optional)
    parser.add_argument("--chains", help="Number of chains (default: as pm.sample -- max(2, cores), cores = --cores or "
                        "min(4, CPU count)).", type=int)
    parser.add_argument("--seed", help="Random seed.", type=int, default=0)
    parser.add_argument("--device", help="HIP device ordinal.", type=int, default=-1)
    parser.add_argument("--no_deterministics", help="Do not record i / ab_n_mu / ab_s_mu per draw.", action="store_true")
    parser.add_argument("--no_discrete", help="Do not record i_raw / ab_s_waner per draw.", action="store_true")
    parser.add_argument("--thin", "--record_every", dest="thin", type=int, default=1,
                        help="Keep the per-draw (gap, ind) arrays of every K-th draw only (the 17 scalars, the statistics and the "
                        "posterior means mean_i / mean_ab_n_mu / mean_ab_s_mu cover every draw).  Default 1; a run whose arrays "
                        "exceed the host budget (ABD_RECORD_BUDGET_GB, default 8) is refused with the K that fits.")
    parser.add_argument("--dense_metric", help="Adapt a full mass matrix (PyMC's init='adapt_full') instead of a diagonal one.",
                        action="store_true")
    return parser


def write_posterior(res: dict, path: str, coords: dict) -> str:
    try:
        import arviz as az  # noqa: F401
    except ImportError:
        az = None
    if az is not None and path:
        # (UNVERIFIED-OFFLINE: ArviZ is not importable where this was written)
        dims = {"i_raw": ["gap", "ind"], "i": ["gap", "ind"], "ab_n_mu": ["gap", "ind"], "ab_s_mu": ["gap", "ind"],
                "ab_s_waner": ["ind"], "mean_i": ["chain", "gap", "ind"], "mean_ab_n_mu": ["chain", "gap", "ind"],
                "mean_ab_s_mu": ["chain", "gap", "ind"]}
        skip = ("n_grad_evals", "draw_index", "mean_i", "mean_ab_n_mu", "mean_ab_s_mu")
        post = {k: v for k, v in res.items() if not k.startswith("stat_") and k not in skip}
        stats = {k[5:]: v for k, v in res.items() if k.startswith("stat_")}
        means = {k: res[k] for k in ("mean_i", "mean_ab_n_mu", "mean_ab_s_mu") if k in res}
        if "draw_index" in res and res["draw_index"].shape[1] != next(iter(stats.values())).shape[1]:
            # --thin: an InferenceData has ONE draw axis, so the file holds the thinned draws of every variable (what
            # abdpymc-subsample-idata makes of a full one); the posterior means over ALL draws travel as constant data
            idx = res["draw_index"][0]
            n_all = next(iter(stats.values())).shape[1]
            post = {k: (v[:, idx] if v.shape[1] == n_all else v) for k, v in post.items()}
            stats = {k: v[:, idx] for k, v in stats.items()}
        idata = az.from_dict(posterior=post, sample_stats=stats, constant_data=means or None, coords=coords, dims=dims)
        az.to_netcdf(idata, path)  # abd.py:924
        return path
    out = (path or "abd_posterior") + ("" if str(path or "").endswith(".npz") else ".npz")
    np.savez_compressed(out, **res, coord_gap=coords["gap"], coord_ind=coords["ind"])
    return out


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    from . import distributed
    from .data import TiterData
    from .model import model
    from .sampler import sample

    # one process per GPU under torch.distributed.run: chains are sharded, draws gathered at the end
    dist, rank, world, local = distributed.init_from_env()

    data = TiterData.from_disk(args.ititers_data)  # abd.py:913
    splits = (
        None
        if (not args.split_delta) and (not args.split_omicron)
        else data.calculate_splits(delta=args.split_delta, omicron=args.split_omicron)
    )  # abd.py:915-919
    # pm.sample(cores=...) (abd.py:922) runs chains = max(2, cores) with cores = min(4, CPU count) when not given
    import os

    cores = args.cores or min(4, os.cpu_count() or 1)
    chains = args.chains or max(2, cores)
    counts = distributed.split_counts(chains, world)
    mine, first = counts[rank], sum(counts[:rank])
    if chains < world:
        # every rank sees the same numbers and leaves together (a rank that went on alone would hang in the gather)
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(f"{chains} chains cannot be sharded over {world} processes: fewer chains than ranks")
    device = args.device
    if world > 1 and device < 0:
        import torch

        device = local % max(1, torch.cuda.device_count())
    m = model(data, splits=splits, ignore_pcrpos=args.ignore_pcrpos, n_chains=mine, device=device)  # abd.py:921
    t0 = time.time()

    def progress(c, a, b):
        if a == b or a % max(1, b // 10) == 0:
            print(f"chain {first + c}: {a}/{b} iterations, {time.time() - t0:.1f} s", file=sys.stderr, flush=True)

    if args.thin < 1:
        raise SystemExit(f"--thin must be >= 1, got {args.thin}")
    res = sample(m, tune=args.tune, draws=args.draws, chains=mine, seed=args.seed,
                 record_deterministics=not args.no_deterministics, record_discrete=not args.no_discrete, progress=progress,
                 chain_offset=first, dense_metric=args.dense_metric, thin=args.thin)  # abd.py:922
    name = m.ctx.device_name
    m.close()
    if world > 1:
        import torch

        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else None
        res = distributed.gather_results(res, counts, dist, dev)
    if rank == 0:
        out = write_posterior(res, args.netcdf, data.coords)
        print(f"wrote {out}  ({chains} chains x {args.draws} draws on {world} x {name})", file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
