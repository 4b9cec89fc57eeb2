/*
 * abd_hip.h -- C ABI of the MI355X-native abdpymc joint log-probability hot path.
 *
 * This is the drop-in boundary for ONE path of davipatti/abdpymc: the joint logp (+ gradient) of the
 * antibody-dynamics model built by abdpymc.model() (reference abdpymc/abd.py:396-442) and evaluated by
 * PyMC's two compiled callables inside pm.sample() (reference call site abd.py:922):
 *
 *   Model.compile_logp()            point -> scalar logp          -> abd_logp
 *   Model.logp_dlogp_function()     theta[17] -> (logp, grad[17]) -> abd_logp_dlogp / _batch
 *   pm.Deterministic "i", "ab_n_mu", "ab_s_mu" (abd.py:649/667, 341, 389-391) -> abd_deterministics
 *
 * Plain C: pointers and sizes only.  The caller owns every host buffer passed in or out; the library
 * copies inputs at abd_create / abd_set_discrete and owns all device memory until abd_destroy.
 * Every function returns 0 on success and a negative abd_status on error; the message is available from
 * abd_last_error().  Numerical out-of-range (e.g. exp overflow of a sigma) is NOT an error: logp comes
 * back -inf / nan with status 0, as PyMC signals it (NUTS marks a divergence).
 *
 * theta layout (17 doubles, PyMC value variables in creation order; abd.py:424, 329-340, 367-388, 464-467):
 *   0 p_logodds__          1 ab_n_perm_log__     2 ab_n_temp_log__     3 ab_n_rho_logodds__   4 ab_n_init
 *   5 ab_s_perm_log__      6 ab_s_rho_logodds__  7 ab_s_p_waner_logodds__
 *   8 ab_s_tempinf_log__   9 ab_s_tempvac_log__  10 ab_s_init
 *   11 it_n_b  12 it_n_d  13 it_n_sigma_log__    14 it_s_b  15 it_s_d  16 it_s_sigma_log__
 */
#ifndef ABD_HIP_H
#define ABD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABD_N_THETA 17
/* Limit of this library that the reference does not have: abdpymc takes any n_gaps (abd.py:101, 224-239); here the
 * kernels that hold an individual's gap axis in registers (observation lists, the sweep, the Deterministics) are built
 * for 4 and for 8 packed 64-bit words, so n_gaps <= 512 (abd_create refuses more with ABD_ERR_ARG; the dense evaluation
 * kernel reads words on demand and has no such limit of its own).  The reference's cohorts have 26 / 31 monthly gaps,
 * BASELINE's synthetic ones 60 / 200.  Beyond 256 gaps dense cohorts sweep with the wave-per-proposal kernel. */
#define ABD_MAX_GAPS 512
#define ABD_MAX_BATCH 16  /* chains per kernel launch (larger batches are split) */

typedef enum {
  ABD_OK = 0,
  ABD_ERR_ARG = -1,     /* bad argument (maps to ValueError in the Python mirror)            */
  ABD_ERR_HIP = -2,     /* HIP runtime failure; message carries hipGetErrorString             */
  ABD_ERR_STATE = -3,   /* e.g. logp on a chain slot whose discrete state was never set       */
  ABD_ERR_NOMEM = -4
} abd_status;

typedef enum { ABD_STORE_F64 = 0, ABD_STORE_F32 = 1 } abd_storage;

/* One antigen's observation list = reference AntigenTiterData (abd.py:22-43): OD readings gathered out
 * of the (gap, ind) titer matrix by mu[idx_gap, idx_ind] (abd.py:343, 393).  Any order; the library
 * sorts by (ind, gap) and detects the dense-panel case (exactly one reading in every cell). */
typedef struct {
  int64_t n_obs;
  const int32_t* idx_gap;      /* df.elapsed_months   abd.py:35 */
  const int32_t* idx_ind;      /* df.individual_i     abd.py:36 */
  const double* log_dilution;  /* abd.py:462 */
  const double* od;            /* abd.py:468 */
} abd_antigen_obs;

/* Everything abd.model(data, splits, ignore_pcrpos) closes over (abd.py:396-442). */
typedef struct {
  int32_t n_gaps;              /* G  = TiterData.n_gaps (abd.py:101); Beta prior of p uses it (abd.py:424) */
  int32_t n_inds;              /* N  = TiterData.n_inds (abd.py:102) */
  int32_t n_splits;            /* 0, 1 or 2 (abd.py:865-882) */
  int32_t splits[2];           /* ascending gap indexes, check_splits rules (abd.py:604-622) */
  int32_t storage;             /* abd_storage: precision the OD / log_dilution panels are HELD in on the device */
  int32_t n_chain_slots;       /* independent (i_raw, waner) states resident at once */
  int32_t device;              /* HIP device ordinal; <0 = current device */
  abd_antigen_obs s;           /* measurement '10222020-S' (abd.py:85) */
  abd_antigen_obs n;           /* measurement '40588-V08B' (abd.py:94) */
  const int8_t* vacs;          /* (N, G) row-major 0/1 = TiterData.vacs   (abd.py:114) */
  const int8_t* pcrpos;        /* (N, G) row-major 0/1 = TiterData.pcrpos (abd.py:115); NULL = ignore_pcrpos (abd.py:416-418) */
} abd_desc;

typedef struct abd_ctx abd_ctx;

/* Library identity / build info (static string). */
const char* abd_version(void);

/* Last error message of the calling thread (never NULL). */
const char* abd_last_error(void);

/* Build a context: validates like the reference (same conditions as abd.py:196-197, 604-622),
 * sorts + uploads the observation panels, allocates chain slots.  Lazily initialises HIP on first call
 * in the process (fork-safe: nothing touches the device at load time). */
int abd_create(const abd_desc* desc, abd_ctx** out);
int abd_destroy(abd_ctx* ctx);

/* Device description, e.g. "AMD Instinct MI355X gfx950 256 CUs". */
int abd_device_name(abd_ctx* ctx, char* buf, int32_t buflen);

/* Replace chain slot `chain`'s discrete state: i_raw is (G, N) row-major exactly as PyMC holds the
 * value variable "i_raw" (dims gap, ind; abd.py:427); waner is (N,) = "ab_s_waner" (abd.py:373).
 * Values must be 0/1. */
int abd_set_discrete(abd_ctx* ctx, int32_t chain, const int8_t* i_raw, const int8_t* waner);

/* Flip one bit of the resident discrete state without re-uploading it (what BinaryGibbsMetropolis does
 * between two logp calls).  flat < G*N addresses i_raw.ravel(); flat >= G*N addresses waner[flat-G*N]. */
int abd_flip_discrete(abd_ctx* ctx, int32_t chain, int64_t flat);

/* Read the resident discrete state of `chain` back: i_raw (G, N) and waner (N,), either may be NULL. */
int abd_get_discrete(abd_ctx* ctx, int32_t chain, int8_t* i_raw, int8_t* waner);

/* One binary Gibbs-Metropolis sweep over [i_raw, ab_s_waner] of each listed chain, in place, at theta
 * (n x 17): what PyMC's BinaryGibbsMetropolis.astep does to these two variables inside pm.sample
 * (abd.py:922) -- every dim proposed with probability 0.8 in a uniformly random order, Metropolis
 * acceptance on the joint logp -- using the fact that a flip only changes its own individual's terms.
 * Randomness is Philox4x32-10 keyed by (seed, sweep) with the chain's slot id in the counter: a chain's sweep
 * does not depend on which other chains are listed.  accepted / proposed (n each) may be NULL. */
int abd_gibbs_sweep(abd_ctx* ctx, int32_t n, const int32_t* chains, const double* theta, uint64_t seed,
                    uint32_t sweep, int64_t* accepted, int64_t* proposed);

/* Scalar joint logp at (theta, resident discrete state of `chain`).  Replaces Model.compile_logp()'s
 * point function (a17). */
int abd_logp(abd_ctx* ctx, int32_t chain, const double* theta, double* logp);

/* logp and d logp / d theta.  Replaces Model.logp_dlogp_function() (a18). */
int abd_logp_dlogp(abd_ctx* ctx, int32_t chain, const double* theta, double* logp, double* grad);

/* Only the part of the joint logp that reads the OD panels -- the two observed Normal terms "it_n_lik",
 * "it_s_lik" (abd.py:459-469) -- and its gradient w.r.t. theta (entries the data term does not depend on
 * are 0).  For callers that keep the priors in PyMC and attach this as a pm.Potential. */
int abd_loglik_dlogp(abd_ctx* ctx, int32_t chain, const double* theta, double* loglik, double* grad);

/* n evaluations in as few launches as possible: chains[k] in [0, n_chain_slots), theta is n x 17,
 * logp n, grad n x 17.  The shared OD panels are read once per launch for all chains in it. */
int abd_logp_dlogp_batch(abd_ctx* ctx, int32_t n, const int32_t* chains, const double* theta,
                         double* logp, double* grad);

/* Stream-ordered form: enqueue returns as soon as the launch is queued; results land in result slot
 * `slot` (0 <= slot < abd_n_result_slots) and are read back with abd_fetch after abd_wait.
 * A NUTS driver that runs several chain groups uses this to overlap host work with the device.
 * Dense cohorts: consecutive enqueued launches rotate over up to four HIP streams that sit on different hardware
 * queues (measured at abd_create; launch k+4 sums launch k's partials), each with a quarter of the workgroups of a
 * synchronous launch, so four share the chip instead of one draining it between launches; abd_wait joins them.
 * Synchronous calls may be interleaved: they use rows of their own and leave every result slot alone.
 * Each form is bit-reproducible; the two forms use different launch shapes and agree to rounding (~1e-15). */
int abd_n_result_slots(abd_ctx* ctx);
int abd_logp_dlogp_batch_enqueue(abd_ctx* ctx, int32_t slot, int32_t n, const int32_t* chains,
                                 const double* theta);
int abd_wait(abd_ctx* ctx);
int abd_fetch(abd_ctx* ctx, int32_t slot, double* logp, double* grad);
/* abd_fetch for several slots in one call; outputs are concatenated in the order of `slots`. */
int abd_fetch_many(abd_ctx* ctx, int32_t n_slots, const int32_t* slots, double* logp, double* grad);
/* n_steps independent evaluations of the same chains in one call: theta is n_steps x n x 17, logp n_steps x n, grad
 * n_steps x n x 17 (NULL: logp only).  The stream-ordered form end to end -- enqueue every step, wait once, fetch --
 * for callers that hold many points at once (tempering, particle methods, a benchmark); result slots 0 .. are used. */
int abd_logp_dlogp_many(abd_ctx* ctx, int32_t n_steps, int32_t n, const int32_t* chains, const double* theta,
                        double* logp, double* grad);

/* The three recorded Deterministics for chain slot `chain` at theta, each (G, N) row-major as PyMC
 * stores them (dims gap, ind).  Any output pointer may be NULL. */
int abd_deterministics(abd_ctx* ctx, int32_t chain, const double* theta, int8_t* i, double* ab_n_mu,
                       double* ab_s_mu);

/* ---------------------------------------------------------------------------------------------------
 * The compound step pm.sample assigns to this model (reference call site abd.py:921-922), run natively
 * for several chains: NUTS on the 17 continuous variables, then one Gibbs sweep of [i_raw, ab_s_waner], then a
 * re-evaluation at the new discrete state.  This removes the host-language cost per leapfrog (PyMC: Python;
 * SURVEY 8f rank 4).  The chains run as independent units of 1-8 consecutive chains, each on its own HIP stream
 * with its own result rows: a unit's pending leapfrog points go out as ONE evaluation launch, its sweep and the
 * re-evaluation are queued behind each other on its stream, and the host polls completion tags -- no unit waits
 * for another one's trees (as PyMC's one process per chain does not), and their launches overlap on the device.
 * What a unit computes depends only on the unit (fixed launch shape), never on the other units or on timing.
 * Leapfrog trains (diagonal metric; ABD_SAMPLER_TRAINS): the launch that evaluates a point also assembles logp and gradient,
 * finishes the leapfrog and leaves the next point in device memory for the launch the host has already queued behind it,
 * so a chain's leapfrogs follow each other at the device's pace, not at the host's round trip.
 *   - Dense cohorts (abd_train.hpp): every chain has a small state machine in device memory -- the two ends of its tree, the
 *     doubling directions of the transition (drawn by the host before the tree starts), the leaves still to go.  A launch takes
 *     the 1, 2 or 4 chains of its unit one leapfrog further each, whatever stage each of them is in, and goes on ACROSS the
 *     halves of a doubling and across doublings; the host reads the launches' records from a ring in mapped memory, runs the
 *     tree logic (U-turn tests, multinomial choice) behind the device and tells the state machine when a transition is over
 *     (launches queued beyond that point find the chain idle and skip it).  The chain's sweep runs on a side stream of its
 *     own while the other chains of the unit go on.  Units: one chain up to 4 chains, two up to 7, four beyond.
 *   - Observation lists: units of one chain; a train ends with the half of the doubling it serves.
 * A train run is exactly repeatable, does not depend on the other chains or on timing, and equals the host-driven run
 * (ABD_SAMPLER_TRAINS=0) to rounding, not bit for bit: the closed forms of the transforms are the device's exp / log1p
 * there.
 * Cohorts kept as observation lists are bound by the host's two kernel launches per evaluation, so their units are
 * driven by T host threads inside abd_sampler_run (T a power of two <= 8, 4 by default; thread t the units u with
 * u mod T == t: what a thread touches is private to its units); dense cohorts are bound by the device and use the calling
 * thread only.  abd_kernel_timing and the threaded sampler are mutually exclusive: while a timing mode is on, the units
 * are driven by the calling thread alone.
 * Step size: dual averaging to `target_accept`; metric: diagonal, windowed running variance of the tuning draws
 * (abdpymc_amd/csrc/abd_nuts.hpp).
 * Iterations [0, tune) adapt; later ones are draws.  Randomness: one xoshiro256++ stream per chain keyed by
 * (seed, chain slot) for NUTS; Philox keyed by (seed, iteration) for the sweep. */
typedef struct abd_sampler abd_sampler;

typedef struct abd_sampler_opts {
  int64_t tune;           /* iterations that adapt step size and metric */
  uint64_t seed;
  double target_accept;   /* 0.8 as pm.sample */
  int32_t max_treedepth;  /* 10 as pm.sample; at most 16 */
  int32_t gibbs;          /* 1: sweep [i_raw, ab_s_waner] after every NUTS transition; 0: continuous part only */
  int32_t accumulate;     /* 1: add i, ab_n_mu, ab_s_mu of every draw (iteration >= tune) into device sums */
  int32_t chain_offset;   /* chain slot k draws from random stream k + chain_offset: the global chain id when chains
                             are sharded over processes (one per GPU), so draws do not depend on the world size */
  int32_t dense_metric;   /* 0: diagonal M^-1, pm.sample's default; 1: full covariance of the tuning draws (PyMC:
                             init="adapt_full") -- this posterior is strongly correlated and trees get ~8x shorter */
  int32_t reserved;
} abd_sampler_opts;

#define ABD_N_STATS 11
/* columns of the per-iteration statistics row */
#define ABD_STAT_LP 0                /* joint logp after the whole compound step */
#define ABD_STAT_TREE_DEPTH 1
#define ABD_STAT_N_STEPS 2           /* leapfrogs = logp_dlogp evaluations of the transition */
#define ABD_STAT_MEAN_TREE_ACCEPT 3
#define ABD_STAT_STEP_SIZE 4
#define ABD_STAT_DIVERGING 5
#define ABD_STAT_ENERGY 6
#define ABD_STAT_MAX_ENERGY_ERROR 7
#define ABD_STAT_GIBBS_ACCEPTED 8
#define ABD_STAT_GIBBS_PROPOSED 9
#define ABD_STAT_T_DONE 10           /* seconds from the start of this abd_sampler_run call to the end of the chain's iteration
                                        (host clock): chains are independent, so they finish their iterations at different times */

/* chains[k] must hold a discrete state (abd_set_discrete); theta0 is n x 17, the starting points.
 * A sampler points into its context: destroy it before abd_destroy(ctx). */
int abd_sampler_create(abd_ctx* ctx, int32_t n, const int32_t* chains, const double* theta0,
                       const abd_sampler_opts* opts, abd_sampler** out);
void abd_sampler_destroy(abd_sampler* s);
/* Advance every chain by n_iter iterations.  theta: n x n_iter x 17, stats: n x n_iter x ABD_N_STATS
 * (either may be NULL).  The discrete state after the call is read with abd_get_discrete. */
int abd_sampler_run(abd_sampler* s, int64_t n_iter, double* theta, double* stats);

/* abd_sampler_run that also records, for every iteration of the call, the discrete state and the three
 * Deterministics of every chain -- what the reference keeps per draw in its InferenceData (abd.py:427, 373,
 * 649/667, 341, 389-391).  Each array is [n][capacity][...] (chain-major; (G, N) or (N,) per draw, as PyMC stores
 * them) and this call fills draws first .. first + n_iter - 1 of every chain; NULL arrays are skipped.  The
 * draws are staged on the device and copied out in large blocks (at config 3 a draw is 36 MB per chain). */
typedef struct abd_record {
  int64_t capacity;
  int64_t first;
  int64_t thin;        /* 0 or 1: every iteration of the call is recorded; K > 1: iterations 0, K, 2K, ... of the call, at
                          draws first, first + 1, ... (the reference thins afterwards, subsample_idata.py; at config 3 a
                          draw is 36 MB per chain, so here it is done while sampling) */
  int8_t* i_raw;
  int8_t* ab_s_waner;
  int8_t* i;
  double* ab_n_mu;
  double* ab_s_mu;
} abd_record;
int abd_sampler_run_record(abd_sampler* s, int64_t n_iter, double* theta, double* stats, const abd_record* rec);
/* Posterior means of the Deterministics of chain k (0 <= k < n) over the draws accumulated so far, each
 * (G, N); any pointer may be NULL.  *n_draws receives the number of accumulated draws. */
int abd_sampler_means(abd_sampler* s, int32_t k, double* i_mean, double* ab_n_mu_mean, double* ab_s_mu_mean,
                      int64_t* n_draws);
/* Current diagonal of M^-1 (17) and step size of chain k; `metric` (17 x 17, may be NULL) receives the full
 * M^-1 (the diagonal matrix when the metric is diagonal). */
int abd_sampler_adaptation(abd_sampler* s, int32_t k, double* inv_mass, double* step_size, double* metric);
/* Install a diagonal M^-1 (17 entries, all > 0; NULL keeps the chain's) and / or a step size (<= 0 keeps the chain's) for chain
 * k between two abd_sampler_run calls, e.g. one adaptation pooled over the chains (PyMC: pm.sample(step=pm.NUTS(scaling=...,
 * step_scale=...))).  During iterations < tune the chain goes on adapting from there. */
int abd_sampler_set_adaptation(abd_sampler* s, int32_t k, const double* inv_mass, double step_size);

/* ---------------------------------------------------------------------------------------------------
 * One chain over several GPUs (cohorts too large or too slow for one): the joint logp is a sum over
 * individuals plus terms of theta alone, so each process holds a context over ITS slice of the individuals and
 *   logp(theta) = sum over processes of abd_logp_dlogp(...)  -  (processes - 1) x abd_theta_prior(theta)
 * (likewise the gradient): one all-reduce of 18 doubles per evaluation.  The Gibbs sweep needs no exchange at all;
 * abd_set_individual_offset makes it draw the random numbers of the individual's GLOBAL index, so the sharded
 * sweep is bit-identical to the unsharded one.  Python: abdpymc_amd.distributed.IndividualShards. */
/* The part of the joint logp that depends on theta only (continuous priors + log-Jacobians; no Bernoulli
 * term, no data term) and its gradient (may be NULL). */
int abd_theta_prior(abd_ctx* ctx, const double* theta, double* logp, double* grad);
int abd_set_individual_offset(abd_ctx* ctx, int64_t first_individual);

/* Measurement hooks used by bench.py.  mode 1: every evaluation kernel launch is bracketed by HIP events on
 * the stream it is launched on, and stream-ordered launches all go to ONE stream with the full grid (normally
 * they rotate over four, so that launches share the chip: a launch's own duration is only meaningful when
 * nothing else is in flight) -- the isolated kernel.  mode 2: the launch shape is left alone and HIP events
 * bracket every WINDOW of stream-ordered launches (first abd_logp_dlogp_batch_enqueue after an abd_wait ..
 * every stream joined at the next abd_wait) -- device time per launch as a stream-ordered caller runs them.
 * mode 0: off.  abd_kernel_time returns the accumulated device time and launch count since the last reset
 * (synchronises). */
int abd_kernel_timing(abd_ctx* ctx, int32_t mode);
/* Every waiting call (synchronous evaluations, abd_wait, abd_logp_dlogp_many, the native sampler) waits for its result
 * rows by polling a completion tag in mapped host memory; if a tag does not show in time (~2 M polls / 1 s) the call
 * falls back to a stream synchronise, checks the tags again and fails with ABD_ERR_STATE if a row still lacks its tag.
 * Number of such fall-backs since abd_create: anything but 0 means the tag path has regressed. */
int64_t abd_wait_fallbacks(abd_ctx* ctx);
/* HIP multiplexes its streams over a few hardware queues, and kernels of streams that share a queue run one after the
 * other.  Measures (once per context, ~0.5 ms) which of the context's n <= 8 streams share one: streams with the same
 * number in queue_of_stream[] do.  The native sampler gives its units streams of different queues first. */
int abd_stream_queues(abd_ctx* ctx, int32_t* queue_of_stream, int32_t n);
int abd_kernel_time(abd_ctx* ctx, double* total_ms, int64_t* launches, int32_t reset);

/* Tuning hook (benchmarks / experiments): number of 256-thread workgroups of the evaluation grid
 * (<= 0 keeps the current value) and chains evaluated per wavefront (0 = automatic, else 1, 2 or 4). */
int abd_set_launch_config(abd_ctx* ctx, int32_t blocks, int32_t chains_per_wave);


/* Compulsory bytes one launch of `n_chains` evaluations has to read in THIS library's device layout:
 * dense: G*N*4R (the two [od, log_dilution] panels) + bit-packed indicator words
 * (vacs, pcrpos, one i_raw per chain: 8 bytes per individual per 64 gaps each) + n_chains*N waner bytes.
 * (SURVEY 8d's byte-per-cell figure, G*N*(4R + 2 + n_chains) + n_chains*N, is larger.) */
int64_t abd_algorithmic_bytes(abd_ctx* ctx, int32_t n_chains);

/* 1 if the observation panels were recognised as dense (one S and one N reading in every cell). */
int abd_is_dense(abd_ctx* ctx);
/* HIP streams that stream-ordered dense launches rotate over (1 for cohorts kept as observation lists). */
int abd_n_pipes(abd_ctx* ctx);

/* ---------------------------------------------------------------------------------------------------
 * Environment variables the library reads (all optional; read at abd_create / abd_sampler_create / first use).
 * This table is complete: the product library calls getenv for nothing else (development knobs of launch shapes
 * exist only in a tuning build compiled with -DABD_TUNING, tools/README.md).
 *
 *   variable             default             meaning
 *   ABD_PIPES            4                   HIP streams (one per hardware queue) that stream-ordered dense launches
 *                                            rotate over; 1 = every launch alone on the context's stream (profiling)
 *   ABD_OBS_LANES        by list density     observation lists: 1 = lane-per-observation kernel, 0 = wave-per-individual
 *   ABD_FORCE_SPARSE     0                   1 = keep a dense panel as observation lists (exercises the list kernels)
 *   ABD_GIBBS_V1         0                   1 = dense cohorts sweep with the wave-per-proposal kernel (cross-check)
 *   ABD_DENSE_OWN_SUM    1                   0 = a sampler unit's launch is summed by a second launch (same bits; no leapfrog trains then)
 *   ABD_SAMPLER_THREADS  1 dense / 4 lists   host threads that drive the native sampler's units (<= 8 are used)
 *   ABD_SAMPLER_UNIT     by cohort           chains per independent unit of the native sampler (dense: 1 up to 4 chains, 2 up to 7,
 *                                            4 beyond; 1, 2 or 4)
 *   ABD_SAMPLER_TRAINS   1                   0 = no leapfrog trains: the host sees every leapfrog before the next is queued
 *   ABD_SAMPLER_PROFILE  0                   1 = abd_sampler_run reports on stderr where the host thread's time went
 *   ABD_GIBBS_STATS      0                   1 = abd_gibbs_sweep reports the dense sweep's scheduler counters on stderr
 * (The Python layer adds ABD_HIP_LIB, the path of this library, and ABD_RECORD_BUDGET_GB, the host memory a process may spend
 * on per-draw (G, N) records; bench.py adds ABD_DIST_BACKEND.  The HIP runtime's HIP_FORCE_DEV_KERNARG must stay at its
 * default of 1: kernel arguments in host memory cost 3-5 us per launch.) */

#ifdef __cplusplus
}
#endif
#endif /* ABD_HIP_H */
