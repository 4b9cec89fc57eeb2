// abd_host.hpp -- host-side internals shared by the translation units of libabd_hip.so:
//   abd_context.hip  context life cycle, discrete state, Deterministics, the closed-form host terms (priors, Jacobians)
//   abd_eval.hip     evaluation launches (dense panels / observation lists), pipes, completion tags, timing
//   abd_gibbs.hip    the device Gibbs sweep
//   abd_sampler.hip  the native compound sampler (NUTS state machine in abd_nuts.hpp + the sweep)
// Nothing here is part of the C ABI (include/abd_hip.h).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/abd_hip.h"
#include "abd_types.hpp"
#include "abd_terms.hpp"

static_assert(ABD_MAX_BATCH == ABD_MAX_BATCH_K, "header / kernel batch size mismatch");
static_assert(ABD_MAX_GAPS == 64 * ABD_MAXT_MAX, "header / kernel gap limit mismatch");
static_assert(ABD_N_THETA == ABD_NT, "header / kernel value-variable count mismatch");

namespace abdi {

int fail(int code, const char* fmt, ...);
const char* last_error();
void set_error(const std::string& msg);

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(ABD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

// The product library reads the few documented ABD_* environment variables of include/abd_hip.h (env_int) and nothing else.
// Development knobs (launch shapes, scheduler constants of the sweep) exist only in a tuning build, -DABD_TUNING
// (tools/README.md): in the product they are the compile-time defaults below.
inline int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
inline int tune_int(const char* name, int dflt) {
#ifdef ABD_TUNING
  return env_int(name, dflt);
#else
  (void)name;
  return dflt;
#endif
}

constexpr int kResultSlots = 1024;
constexpr int kSyncSlot = kResultSlots;  // private rows of the synchronous calls: they never touch a caller's slot
constexpr int kMaxPipes = 8;             // HIP streams of a context
constexpr int kMinRows = 4;

// ABD_SAMPLER_PROFILE: time the host spends inside hipLaunchKernelGGL for evaluation launches and their sums
struct LaunchProfile {
  bool on = false;
  double eval_s = 0.0, sum_s = 0.0;
  long evals = 0, sums = 0;
};
extern LaunchProfile g_launch_profile;

struct AntigenDev {
  int64_t K = 0;
  void* y = nullptr;       // sparse: R[K], sorted by (ind, gap)
  void* x = nullptr;       // sparse: R[K]
  uint16_t* g = nullptr;   // sparse: gap per obs
  int32_t* ptr = nullptr;  // sparse: (N+1)
  int32_t* j = nullptr;    // sparse: individual per obs
  void* yx = nullptr;      // dense: [G][N] of {od, log_dilution}
  void* yxi = nullptr;     // dense: the same pairs individual-major, [N][G] (the sweep kernels' reads)
  // dense, when the antigen has <= 256 distinct log dilutions: the split panels of one-chain launches (abd_dense.hpp: XC)
  void* od = nullptr;        // [lane group][G][64] od in the storage type (lane-group-major)
  uint8_t* xc = nullptr;     // [lane group][G][64] code of the cell's log dilution
  double* dict = nullptr;    // [n_dict] the distinct log dilutions
  int n_dict = 0;
};

struct ChainSlot {
  uint64_t* rw = nullptr;   // [nt][N] packed i_raw
  int8_t* waner = nullptr;  // [N]
  // derived from (rw, waner) and kept current by every writer of the slot (abd_set_discrete, abd_flip_discrete, the sweep
  // kernels): the constrained infections the evaluation kernels read, and {sum(i_raw), sum(ab_s_waner)}
  uint64_t* iw = nullptr;      // [nt][N] packed i = constrain(i_raw, pcrpos)   abd.py:640-667
  long long* cnt = nullptr;    // [2]
  bool set = false;
};

struct ResultSlot {
  int n = 0;
  bool grad = true;
  std::vector<int32_t> chains;
  std::vector<double> theta;  // n x 17
  std::vector<HostTerms> host;  // n: the host-side terms of every theta, computed when the evaluation is queued
  double tag_first = 0.0;       // completion tag of the slot's first group of <= ABD_MAX_BATCH rows (group g: tag_first + g)
};

}  // namespace abdi

using namespace abdi;

struct abd_ctx {
  int device = 0;
  int G = 0, N = 0, nt = 0, n_chunks = 1;
  int storage = ABD_STORE_F64;
  bool dense = false;
  bool xc_ok = false;  // dense and both antigens have split panels
  int xc_max_cb = 1;   // launches with at most this many chains per workgroup read them
  bool ignore_pcr = false;
  int n_slots = 0;
  int n_cu = 256;
  int n_lg = 0;           // 64-individual lane groups
  int blocks_x = 0;       // sparse kernel grid (wave per individual)
  int ob_n = 0, ob_s = 0, ob_c = 0;  // observation-lane kernel: workgroups per segment
  bool obs_lanes = true;  // sparse lists: lane per observation (abd_obs.hpp) or wave per individual
  int blocks_max = 0;     // rows per chain in `partials`
  int cpw_forced = 0;
  int dense_blocks = 0;   // dense kernel grid.x
  uint64_t chunk_mask[3][ABD_MAXT_MAX] = {};
  AntigenDev s, n;
  uint64_t* vw = nullptr;  // [nt][N]
  uint64_t* pw = nullptr;  // [nt][N]
  double* exp2_tab = nullptr;  // dense cohorts: 2^(j/1024) (abd_dense.hpp)
  int8_t* stage_gn = nullptr;  // (G, N) upload staging for i_raw
  std::vector<ChainSlot> slots;
  // A pipe = a HIP stream with its own pair of partial buffers and its own pending fixed-order sum.  Pipe 0 is
  // the context's stream (everything synchronous runs there).  Stream-ordered dense launches alternate between
  // pipe 0 and pipe 1: launch k+2 sums launch k's partials (same pipe), so the two streams never wait for each
  // other and the head of one launch overlaps the tail of the previous one.
  struct Pipe {
    hipStream_t st = nullptr;
    double* partials[2] = {nullptr, nullptr};  // [n_slots][blocks_max][ABD_NOUT], alternating per launch
    int pbuf = 0;
    bool on = false;  // pending: the fixed-order sum of the last launch's partials has not been queued yet
    int buf = 0, n = 0, blocks = 0;
    double* out = nullptr;
    double tag = 0.0;
    bool busy = false;  // pipe 1: work queued since the last join with pipe 0
  } pipe[kMaxPipes];
  int n_pipes = 4;        // streams that stream-ordered dense launches rotate over (1 = everything on the context's stream); at most one per hardware queue
  int n_streams = kMaxPipes;  // pipes that exist (the native sampler gives every chain a stream: chain k -> pipe k mod 8)
  int n_sync_slots = 4;   // private result rows of synchronous calls (slot kSyncSlot) and of the sampler's chains in flight
  int pipe_blocks = 0;    // dense grid of a launch that shares the chip with n_pipes - 1 others
  int group_blocks = 0;   // dense grid of one of the sampler's chain groups in flight (set by abd_sampler_create)
  int dbpc = 4;           // dense kernel: workgroups per CU of a launch that has the chip to itself
  int next_pipe = 0;
  hipEvent_t join_ev[kMaxPipes] = {};
  double seq = 0.0;  // completion tags: 1, 2, 3, ... (exact in a double)
  bool fuse_finalize = true;
  // A sampler unit's dense launch sums its own partial rows (abd_dense.hpp; ABD_DENSE_OWN_SUM=0: second launch).  Same
  // bits; the result arrives 2-2.7 us later than from the pre-queued second launch, but the host spends 3.6 instead of
  // 7.2 us per result: config 3, evaluations/s seen by NUTS 33.6 k -> 36.9 k (1 chain), 53 k -> 59 k (8), 77 k -> 88 k (16),
  // unchanged with 4
  bool dense_own_sum = true;
  unsigned int* d_fin_count = nullptr;  // [kMaxPipes][ABD_MAX_BATCH] zeroed counters of that sum
  unsigned int* d_train_count = nullptr;  // [kMaxPipes][ABD_MAX_BATCH][1 + ABD_TRAIN_SHARDS][ABD_TRAIN_CNT_STRIDE] zeroed counters of the two-level count-in (abd_dense.hpp: train launches use row 0 of their pipe, synchronous calls one row per grid row)
  bool sync_own_sum = false;  // tuning build only: a synchronous call's dense launch sums its own partial rows (measured slower than the second launch)
  uint32_t ind_offset = 0;  // global index of this context's first individual (Gibbs random streams)
  bool xcd_remap = true;
  int fin_rows = 2;
  int steps_behind = -1;  // abd_logp_dlogp_many: steps still to be queued behind the one being queued (-1: unknown)
  double prior_const = 0.0;
  double* h_out = nullptr;     // pinned + mapped: [kResultSlots + n_sync_slots][n_slots][ABD_NOUT]
  double* d_out = nullptr;     // device view of h_out
  std::vector<int> pending_slots;  // slots queued stream-ordered since the last abd_wait
  unsigned long long* d_counts = nullptr;  // [n_slots][2] Gibbs accepted / proposed
  unsigned int* d_work = nullptr;          // [2][n_slots] work queue heads of abd_gibbs_dense_kernel (second half: per-chain sweeps of the sampler)
  unsigned long long* d_counts_chain = nullptr;  // [n_slots][2] counts of the sampler's per-chain sweeps ...
  unsigned long long* h_counts_chain = nullptr;  // ... and their pinned host copy
  bool gibbs_v1 = false;                   // ABD_GIBBS_V1=1: dense cohorts use the wave-per-proposal kernel too
  int g2_refill_min = ABD_G2_REFILL_MIN, g2_tail_lanes = ABD_G2_TAIL_LANES, g2_tail_age = ABD_G2_TAIL_AGE;  // scheduler knobs of abd_gibbs_dense_kernel (ABD_G2_*)
  double* d_det = nullptr;                 // staging of abd_deterministics: mu_n, mu_s (G*N doubles each), i (G*N bytes)
  std::vector<ResultSlot> results;
  hipStream_t stream = nullptr;
  // timing: 1 = HIP events around every evaluation-kernel launch, launches serialised on one stream with the full
  // grid (the isolated kernel); 2 = HIP events around every WINDOW of stream-ordered launches (first launch after an
  // abd_wait .. all pipes joined at the next abd_wait): the launch shape a stream-ordered caller really runs
  int timing = 0;
  bool win_open = false;
  int64_t win_launches = 0;  // launches inside the windows collected so far
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<hipEvent_t> win_end;  // timing 2: [window][kMaxPipes] end of each pipe's work in the window (the window ends with the latest)
  std::vector<uint32_t> win_mask;   // ... and which pipes had work in it
  size_t ev_used = 0;
  double ev_total_ms = 0.0;
  int64_t ev_count = 0;
  int queue_of_pipe[kMaxPipes] = {};  // probe_stream_queues: streams with the same number share a hardware queue
  int n_queues = 0;                   // 0 = not probed yet
  int pipe_order[kMaxPipes] = {0, 1, 2, 3, 4, 5, 6, 7};  // one stream of every hardware queue first (probe_stream_queues)
  std::vector<double> unit_seq;  // completion-tag sequences of the native sampler's units (abd_sampler.hip: sampler_run_units)
  int64_t wait_fallbacks = 0;  // synchronous calls whose completion tag never showed and that fell back to a stream synchronise
  char name[256] = {0};
};

namespace abdi {

// ---- abd_context.hip
ModelSizes model_sizes(const abd_ctx* c);
void assemble(const abd_ctx* c, const HostTerms& h, const double* t, const double* sums, double* logp, double* grad, bool with_priors = true);
ChainPar chain_par(const abd_ctx* c, int chain, const Transformed& tr);
ChainPar chain_par(const abd_ctx* c, int chain, const double* t);
void base_args(const abd_ctx* c, EvalArgs& a);
#ifdef ABD_STAMPS
unsigned long long* stamps_buffer();  // diagnostic build: in-kernel s_memrealtime stamps (tools/probe_stamps.py)
#endif
int probe_stream_queues(abd_ctx* c);
inline int unit_pipe(const abd_ctx* c, int u) { return c->pipe_order[u % c->n_streams]; }
int check_chains(abd_ctx* c, int n, const int32_t* chains);
// Deterministics of chain `chain` at theta on stream st: written (G, N) gap-major and / or added to running sums
int launch_deterministics(abd_ctx* c, int chain, const double* theta, hipStream_t st, int8_t* out_i, double* out_mun, double* out_mus, double* sums);
// the chain's packed i_raw as (G, N) int8 on stream st
int launch_unpack(abd_ctx* c, int chain, int8_t* dst, hipStream_t st);

// ---- abd_eval.hip
// dynamic LDS of a dense launch with cpw chains per workgroup: power tables, block reduction, 2^(j/1024) table
size_t dense_lds_bytes(int G, int cpw);
int flush_pipe(abd_ctx* c, int pi);
int join_pipes(abd_ctx* c);
int flush_pending(abd_ctx* c);
int flush_ring(abd_ctx* c);
int wait_rows(abd_ctx* c, int slot, int n, double tag, hipStream_t st = nullptr);
int enqueue_slot(abd_ctx* c, int slot, int n, const int32_t* chains, const double* theta, bool grad, bool deferred = false,
                 int force_pipe = -1, double* seqp = nullptr, bool sync_call = false);
int fetch_slot(abd_ctx* c, int slot, double* logp, double* grad, bool with_priors = true);
int enqueue_train_launch(abd_ctx* c, int chain, int pi, TrainArgs* t, const HostTerms& first_terms, double* seqp = nullptr);
int enqueue_dense_train(abd_ctx* c, int pi, int cb, int blocks, DenseTrainArgs* a);
int dense_blocks(const abd_ctx* c, int cpw, int share = 0, int grid_rows = 1);

// ---- abd_gibbs.hip
int enqueue_gibbs(abd_ctx* c, int m, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
                  uint32_t stream_offset, hipStream_t st, unsigned long long* counts_dev, unsigned int* work_dev,
                  unsigned long long* stats_dev);

}  // namespace abdi
