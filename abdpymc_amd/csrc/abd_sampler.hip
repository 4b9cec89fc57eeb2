// abd_sampler.hip -- the native compound sampler of the C ABI (abd_sampler_*; include/abd_hip.h): the step pm.sample
// assigns to this model (abd.py:921-922) -- NUTS on the 17 continuous variables (abd_nuts.hpp), the device Gibbs sweep on
// [i_raw, ab_s_waner], recording of the Deterministics -- run against the evaluation path without leaving the library.
#include "abd_host.hpp"
#include "abd_nuts.hpp"

#include <deque>

struct abd_sampler {
  abd_ctx* c = nullptr;
  int n = 0;
  abd_sampler_opts o{};
  std::vector<int32_t> chains;
  std::vector<abdnuts::AdaptiveNuts> ch;
  int64_t it = 0;
  double* d_sums = nullptr;  // [n][3][G*N]
  int64_t n_accumulated = 0;
  // recording: device staging of up to rec_chunk draws per chain, [n][rec_chunk][...] per variable
  int64_t rec_chunk = 0;
  double* d_rec_mu = nullptr;   // [2][n][rec_chunk][G*N]  (ab_n_mu, ab_s_mu)
  int8_t* d_rec_i8 = nullptr;   // [2][n][rec_chunk][G*N]  (i_raw, i) then [n][rec_chunk][N] (waner)
  std::vector<double> lp, gr;  // starting points' logp / gradient
  int unit = 1;                // chains per independent unit (sampler_run_units)
  int threads = 1;  // host threads that drive the units (sampler_run_units)
  // Leapfrog trains (abd_types.hpp: TrainArgs): dense cohort, one chain per unit, diagonal metric.  Every evaluation of such a
  // sampler is a train launch -- it assembles logp and gradient on the device and leaves the next point of the half for
  // the launch queued behind it -- and the host keeps up to `lookahead` launches of a half queued ahead of the record it
  // is waiting for, so a chain's leapfrogs follow each other at the device's pace, not at the host's round trip.
  bool trains = false;
  bool unit_tags = false;  // the units' launches are tagged from the context's per-unit sequences (several host threads)
  int lookahead = 8;
  static constexpr int kTrainRing = 32;  // records per unit: > lookahead + 1
  struct TrainUnit {
    TrainPoint* slots = nullptr;     // device, [2]
    TrainRecord* rec_h = nullptr;    // mapped host memory, [kTrainRing] ...
    TrainRecord* rec_d = nullptr;    // ... as the device sees it
    uint64_t prod = 0, cons = 0;     // launches queued / records taken (or given up: the rest of a half that ended early)
    double tags[kTrainRing] = {};
    int next_slot = 0;               // the slot the last queued launch leaves its successor's point in
    bool last_own_record = true;     // did the launch queued last write its own record (else its successor passes it on)
    int queued_in_half = 0;          // launches queued for the half that is being built
  };
  std::vector<TrainUnit> tu;
  // Leapfrog trains of dense cohorts (abd_types.hpp: TrainChain; abd_train.hpp): units of 1, 2 or 4 chains share their
  // launches, the device goes on from one half of a tree into the next by itself, the host's tree logic follows behind on
  // the records, and a chain's sweep runs on a stream of its own beside the unit's launches for the other chains.
  bool dtrains = false;
  int dtrain_blocks = 0;     // workgroups (with a range) of a unit's launch: the unit's fixed shape
  int dtrain_lookahead = 3;  // steps of a unit the host keeps queued ahead of the oldest record it has not seen
  struct DChain {
    TrainChain* st = nullptr;        // device
    TrainRecord* ring_h = nullptr;   // mapped host memory [ABD_TRAIN_RING] ...
    TrainRecord* ring_d = nullptr;   // ... as the device sees it
    TrainBegin* begin_h = nullptr;   // mapped host memory [kBeginBlocks]
    TrainBegin* begin_d = nullptr;
    hipStream_t side = nullptr;      // the chain's sweep and its recording kernels
    // mapped host memory: [0], [1] the sweep's accepted / proposed counts, [2] (as a double) the tag of the sweep they belong to
    unsigned long long* done_h = nullptr;
    unsigned long long* done_d = nullptr;
    double sweep_tag = 0.0;          // tag of the chain's last sweep (1, 2, 3, ...)
    int64_t n_rec = 0;               // records the steps queued so far produce (index of the next one)
    int64_t n_begin = 0;             // transitions handed over so far
  };
  static constexpr int kBeginBlocks = 4;
  std::vector<DChain> dc;
};

namespace {

// The end of a chain's sweep as the native sampler's host thread sees it (abd_sampler.hip): the sweep's two counters go to
// mapped host memory and a tag behind them -- polled with a plain memory read (a hipEventQuery per pass of the host's loop
// costs the loop several microseconds for as long as a sweep is running, and every chain's records wait behind it)
__global__ void abd_sweep_done_kernel(const unsigned long long* counts, unsigned long long* host_counts, double* host_tag, double tag) {
  if (threadIdx.x == 0) {
    host_counts[0] = counts[0];
    host_counts[1] = counts[1];
    __threadfence_system();
    __hip_atomic_store(host_tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

void train_free(abd_sampler* s) {
  for (auto& t : s->tu) {
    if (t.slots) (void)hipFree(t.slots);
    if (t.rec_h) (void)hipHostFree(t.rec_h);
  }
  s->tu.clear();
}

int train_alloc(abd_sampler* s) {
  s->tu.resize((size_t)s->n);
  for (auto& t : s->tu) {
    HIP_TRY(hipMalloc(&t.slots, 2 * sizeof(TrainPoint)));
    HIP_TRY(hipMemset(t.slots, 0, 2 * sizeof(TrainPoint)));
    HIP_TRY(hipHostMalloc((void**)&t.rec_h, abd_sampler::kTrainRing * sizeof(TrainRecord), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(t.rec_h, 0, abd_sampler::kTrainRing * sizeof(TrainRecord));
    HIP_TRY(hipHostGetDevicePointer((void**)&t.rec_d, t.rec_h, 0));
  }
  return ABD_OK;
}

void dtrain_free(abd_sampler* s) {
  for (auto& d : s->dc) {
    if (d.st) (void)hipFree(d.st);
    if (d.ring_h) (void)hipHostFree(d.ring_h);
    if (d.begin_h) (void)hipHostFree(d.begin_h);
    if (d.done_h) (void)hipHostFree(d.done_h);
    if (d.side) (void)hipStreamDestroy(d.side);
  }
  s->dc.clear();
}

int dtrain_alloc(abd_sampler* s) {
  s->dc.resize((size_t)s->n);
  for (auto& d : s->dc) {
    HIP_TRY(hipMalloc(&d.st, sizeof(TrainChain)));
    HIP_TRY(hipMemset(d.st, 0, sizeof(TrainChain)));
    HIP_TRY(hipHostMalloc((void**)&d.ring_h, ABD_TRAIN_RING * sizeof(TrainRecord), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(d.ring_h, 0, ABD_TRAIN_RING * sizeof(TrainRecord));
    HIP_TRY(hipHostGetDevicePointer((void**)&d.ring_d, d.ring_h, 0));
    HIP_TRY(hipHostMalloc((void**)&d.begin_h, abd_sampler::kBeginBlocks * sizeof(TrainBegin), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(d.begin_h, 0, abd_sampler::kBeginBlocks * sizeof(TrainBegin));
    HIP_TRY(hipHostGetDevicePointer((void**)&d.begin_d, d.begin_h, 0));
    HIP_TRY(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc((void**)&d.done_h, 4 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(d.done_h, 0, 4 * sizeof(unsigned long long));
    HIP_TRY(hipHostGetDevicePointer((void**)&d.done_d, d.done_h, 0));
  }
  HIP_TRY(hipDeviceSynchronize());
  return ABD_OK;
}

// Queue one launch of unit u's train: the point staged by the host (theta, p_half: the first leapfrog of a half, or a
// plain evaluation with ve = 0), or -- theta == nullptr -- the successor of the launch queued last.
// own_record: nothing is going to be queued behind this launch that could pass its record on (the last leapfrog of a
// half, a plain evaluation, no look-ahead): it writes the record itself.
int train_launch(abd_sampler* s, int u, const double* theta, const double* p_half, double ve, const double* inv_mass, bool own_record) {
  abd_ctx* c = s->c;
  abd_sampler::TrainUnit& t = s->tu[(size_t)u];
  if (t.prod - t.cons >= (uint64_t)abd_sampler::kTrainRing) return fail(ABD_ERR_STATE, "internal: train record ring of unit %d is full", u);
  TrainArgs ta;
  std::memset(&ta, 0, sizeof ta);
  ta.enabled = 1;
  ta.slots = t.slots;
  ta.rec = t.rec_d + (t.prod % abd_sampler::kTrainRing);
  ta.ve = ve;
  ta.prior_const = c->prior_const;
  ta.own_record = own_record ? 1 : 0;
  std::memcpy(ta.inv_mass, inv_mass, sizeof ta.inv_mass);
  HostTerms ht;
  std::memset(&ht, 0, sizeof ht);
  if (theta) {
    ht = prepare(theta);
    ta.use_slot = -1;
    ta.next_slot = 0;
    std::memcpy(ta.first.theta, theta, sizeof ta.first.theta);
    if (p_half) std::memcpy(ta.first.p_half, p_half, sizeof ta.first.p_half);
    std::memcpy(ta.first.tr, &ht.tr, sizeof ta.first.tr);
    std::memcpy(ta.first.L0, ht.L0, sizeof ta.first.L0);
    std::memcpy(ta.first.L1, ht.L1, sizeof ta.first.L1);
  } else {
    ta.use_slot = t.next_slot;
    ta.next_slot = t.next_slot ^ 1;
    if (!t.last_own_record) {  // the predecessor left its record beside the point: this launch passes it on
      const size_t kp = (size_t)((t.prod - 1) % abd_sampler::kTrainRing);
      ta.fwd_rec = t.rec_d + kp;
      ta.fwd_tag = t.tags[kp];
    }
  }
  // (units driven by their own host threads tag their launches from their own sequence: abd_host.hpp, unit_seq)
  if (int rc = enqueue_train_launch(c, s->chains[(size_t)u], unit_pipe(c, u), &ta, ht, s->unit_tags ? &c->unit_seq[(size_t)u] : nullptr)) return rc;
  t.tags[t.prod % abd_sampler::kTrainRing] = ta.tag;
  t.prod += 1;
  t.next_slot = ta.next_slot;
  t.last_own_record = own_record;
  return ABD_OK;
}

// the first leapfrog of the half chain u's tree is about to build, and as many of its successors as the look-ahead allows
int train_begin(abd_sampler* s, int u) {
  abdnuts::Nuts& nu = s->ch[(size_t)u].nuts;
  abd_sampler::TrainUnit& t = s->tu[(size_t)u];
  // a launch whose successor is certain to be queued (any but the half's last, given a look-ahead) leaves its record to it
  const int n_half = nu.half_remaining() + 1;
  auto own = [&](int j) { return s->lookahead == 0 || j == n_half - 1; };
  if (int rc = train_launch(s, u, nu.request(), nu.staged_momentum(), nu.signed_step(), nu.inv_mass, own(0))) return rc;
  t.queued_in_half = 1;
  const int ahead = std::min(s->lookahead, nu.half_remaining());
  for (int k = 0; k < ahead; ++k) {
    if (int rc = train_launch(s, u, nullptr, nullptr, nu.signed_step(), nu.inv_mass, own(t.queued_in_half))) return rc;
    t.queued_in_half += 1;
  }
  return ABD_OK;
}

// has the oldest outstanding record of unit u landed? (never blocks)
bool train_ready(const abd_sampler* s, int u) {
  const abd_sampler::TrainUnit& t = s->tu[(size_t)u];
  if (t.cons >= t.prod) return false;
  const size_t k = (size_t)(t.cons % abd_sampler::kTrainRing);
  if (*(volatile const double*)&t.rec_h[k].tag != t.tags[k]) return false;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return true;
}

// add chain k's Deterministics at its current point to its running sums (stream st)
int accumulate_chain(abd_sampler* s, int k, hipStream_t st) {
  abd_ctx* c = s->c;
  const size_t cells = (size_t)c->G * c->N;
  return launch_deterministics(c, s->chains[(size_t)k], s->ch[(size_t)k].nuts.q, st, nullptr, nullptr, nullptr,
                               s->d_sums + (size_t)k * 3 * cells);
}

}  // namespace

extern "C" {

int abd_sampler_create(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta0, const abd_sampler_opts* opts,
                       abd_sampler** out) {
  if (!c || !chains || !theta0 || !opts || !out) return fail(ABD_ERR_ARG, "NULL argument");
  *out = nullptr;
  int rc = check_chains(c, n, chains);
  if (rc) return rc;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (chains[a] == chains[b]) return fail(ABD_ERR_ARG, "chain %d listed twice", chains[a]);
  if (opts->tune < 0) return fail(ABD_ERR_ARG, "tune=%lld is negative", (long long)opts->tune);
  if (opts->chain_offset < 0) return fail(ABD_ERR_ARG, "chain_offset=%d is negative", opts->chain_offset);
  if (opts->max_treedepth < 1 || opts->max_treedepth > abdnuts::MAX_DEPTH)
    return fail(ABD_ERR_ARG, "max_treedepth=%d outside [1, %d]", opts->max_treedepth, abdnuts::MAX_DEPTH);
  if (!(opts->target_accept > 0.0 && opts->target_accept < 1.0))
    return fail(ABD_ERR_ARG, "target_accept=%g outside (0, 1)", opts->target_accept);
  abd_sampler* s = new (std::nothrow) abd_sampler();
  if (!s) return fail(ABD_ERR_NOMEM, "out of host memory");
  s->c = c;
  s->n = n;
  s->o = *opts;
  s->chains.assign(chains, chains + n);
  s->ch.resize((size_t)n);
  s->lp.resize((size_t)n);
  s->gr.resize((size_t)n * ABD_N_THETA);
  // chains per unit: a large dense cohort keeps the chip busy with one chain per launch and gains most from chains
  // that never wait for each other; a small cohort is bound by the host's ~6 us per launch, which a unit's chains share
  // (measured, tools/probe_nuts_rate.py: config 3 -- 8 chains 88 k evals/s with units of 1, 82 k with 4; 16 chains 92 k / 112 k;
  // default cohort, 16 chains -- 152 k with units of 1, 334 k with 4, 359 k with 8)
  // host threads that drive the units: one for dense cohorts (bound by the device), up to four for observation lists
  // (bound by the host's two launches per evaluation)
  s->threads = c->dense ? 1 : 4;
  s->threads = std::max(1, std::min(16, env_int("ABD_SAMPLER_THREADS", s->threads)));
  // With four host threads (observation lists) the best split is four units -- one per thread and per
  // hardware queue: default cohort, evaluations/s seen by NUTS with 4 / 8 / 16 chains 217 k / 339 k / 491 k against
  // 166 k / 253 k / 300-370 k for the best split on one thread.
  // Large dense cohorts (one host thread): at most about four units -- the hardware queues -- of 1, 2, 4 or 8 chains, the
  // sizes the dense kernel has a shape for (config 3, evaluations/s seen by NUTS over 150-300 iterations: 8 chains 79 k
  // with units of 1, 112 k with 2; 12 chains 97 k / 88 k / 83 k with 2 / 3 / 4; 16 chains 91 k / 100 k with 2 / 4;
  // 32 chains 100 k / 134 k with 4 / 8)
  // With leapfrog trains (one chain per unit) and the streams spread evenly over the hardware queues, eight chains run best
  // as eight units, two per queue: 126 k against 121 k as four units of two; sixteen chains: 145 k as eight units of two,
  // 155 k as four units of four
  const bool trains_ok = (c->dense || c->obs_lanes) && c->dense_own_sum && opts->dense_metric == 0 && env_int("ABD_SAMPLER_TRAINS", 1) != 0;
  int dense_unit = 1;
  while (dense_unit < 8 && 2 * dense_unit <= n / 4) dense_unit *= 2;
  // dense trains: units of 1, 2 or 4 chains (abd_sampler::dtrains)
  s->dtrains = trains_ok && c->dense;
  // as few chains per unit as keep the units within the four hardware queues (a queue runs one kernel at a time): measured at
  // config 3, evaluations/s seen by NUTS while all chains are at work, units of 1 / 2 / 4 chains -- 4 chains 139 k / 138 k /
  // 102 k; 8 chains 140 k / 176 k / 183 k; 16 chains - / 149 k / 195 k
  if (s->dtrains) dense_unit = n <= 4 ? 1 : (n < 8 ? 2 : 4);
  s->unit = (c->dense && ((int64_t)c->G * c->N >= 500000 || s->dtrains)) ? dense_unit : std::max(s->threads > 1 ? 1 : 2, std::min(8, (n + 3) / 4));
  s->unit = env_int("ABD_SAMPLER_UNIT", s->unit);
  s->unit = std::max(1, std::min({s->unit, n, (int)ABD_MAX_BATCH}));
  if (s->dtrains && s->unit != 1 && s->unit != 2 && s->unit != 4) s->dtrains = false;  // (a train unit is a workgroup's waves)

  // several units' launches are in flight: one workgroup per CU each, whatever the number of units -- a unit's numbers
  // must not depend on it
  c->group_blocks = std::min(c->dense_blocks, c->n_cu);
  if (const int gb = tune_int("ABD_GROUP_BLOCKS_PER_CU", 0)) c->group_blocks = std::max(1, std::min(c->n_cu * gb, c->blocks_max));
  // the starting points through the launch shape the units will use
  rc = hipSetDevice(c->device) == hipSuccess ? flush_ring(c) : fail(ABD_ERR_HIP, "hipSetDevice failed");
  if (!rc && (n + s->unit - 1) / s->unit > 1 && tune_int("ABD_PROBE_QUEUES", 1) != 0) rc = probe_stream_queues(c);
  s->trains = trains_ok && !c->dense && s->unit == 1;
  if (s->dtrains) {
    // the unit's launch shape: two workgroups per CU, however many units there are (config 3, evaluations/s seen by NUTS over
    // the call with 1 / 2 per CU: 3 chains 78 k / 78 k but 94 k / 106 k while all are at work, 4 chains 128 k / 129 k and the compound
    // iteration 966 / 1 027 per s, 5 chains 85 k / 120 k, 7 chains 106 k / 123 k, 16 chains as units of four 125 k / 136 k; 3 or 4
    // per CU are slower again: profiles/r04/b_train_grid_*.txt): chains finish their trees and iterations at different times, and
    // a unit that is alone for a while gets through its launches faster on the larger grid.  A unit that is alone for the whole run
    // takes four per CU where its ranges are long enough to pay for the set-up (config 5: 77 gap rows per range)
    const int n_units = (n + s->unit - 1) / s->unit;
    int per_cu = std::min(2, c->dbpc);
    if (n_units == 1) {
      const int64_t rows = (int64_t)c->n_lg * c->G, nsub = ABD_WAVES_PER_BLOCK / std::max(1, s->unit);
      if (rows / ((int64_t)c->n_cu * c->dbpc * nsub) >= 32) per_cu = c->dbpc;
    }
    s->dtrain_blocks = dense_blocks(c, s->unit, 0, 1);                      // (the cap that keeps ranges >= kMinRows rows)
    s->dtrain_blocks = std::min(s->dtrain_blocks, c->n_cu * per_cu);
    if (const int tb = tune_int("ABD_TRAIN_BLOCKS_PER_CU", 0)) s->dtrain_blocks = std::max(1, std::min({c->n_cu * tb, c->blocks_max, dense_blocks(c, s->unit, 0, 1)}));
    s->dtrain_lookahead = std::max(2, std::min(ABD_TRAIN_RING / 2, tune_int("ABD_TRAIN_LOOKAHEAD", 3)));
    if (!rc) rc = dtrain_alloc(s);
  }
  s->lookahead = std::max(0, std::min(abd_sampler::kTrainRing - 2, tune_int("ABD_TRAIN_LOOKAHEAD", 8)));
  if (!rc && s->trains) rc = train_alloc(s);
  for (int u = 0, lo = 0; lo < n && !rc; ++u, lo += s->unit) {
    const int m = std::min(s->unit, n - lo);
    if (s->trains) {  // every evaluation of this sampler is assembled on the device (see abd_sampler::trains)
      const double ones[ABD_N_THETA] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
      rc = train_launch(s, u, theta0 + (size_t)lo * ABD_N_THETA, nullptr, 0.0, ones, true);
      for (long spin = 0; !rc && !train_ready(s, u); ++spin) {
        if (spin > 4000000) {
          __atomic_fetch_add(&c->wait_fallbacks, (int64_t)1, __ATOMIC_RELAXED);
          if (hipStreamSynchronize(c->pipe[unit_pipe(c, u)].st) != hipSuccess) rc = fail(ABD_ERR_HIP, "hipStreamSynchronize failed");
          if (!rc && !train_ready(s, u)) rc = fail(ABD_ERR_STATE, "the record of chain %d's starting point never received its tag", chains[lo]);
          break;
        }
        __builtin_ia32_pause();
      }
      if (!rc) {
        abd_sampler::TrainUnit& t = s->tu[(size_t)u];
        const TrainRecord& r = t.rec_h[t.cons % abd_sampler::kTrainRing];
        s->lp[(size_t)lo] = r.lp;
        std::memcpy(s->gr.data() + (size_t)lo * ABD_N_THETA, r.g, sizeof(double) * ABD_N_THETA);
        t.cons += 1;
      }
      continue;
    }
    rc = enqueue_slot(c, kSyncSlot + u, m, chains + lo, theta0 + (size_t)lo * ABD_N_THETA, true, false, unit_pipe(c, u));
    if (!rc) rc = wait_rows(c, kSyncSlot + u, m, c->seq, c->pipe[unit_pipe(c, u)].st);
    if (!rc) rc = fetch_slot(c, kSyncSlot + u, s->lp.data() + lo, s->gr.data() + (size_t)lo * ABD_N_THETA);
  }
  if (rc) {
    train_free(s);
    dtrain_free(s);
    delete s;
    return rc;
  }
  for (int k = 0; k < n; ++k) {
    if (!std::isfinite(s->lp[(size_t)k])) {
      train_free(s);
      dtrain_free(s);
      delete s;
      return fail(ABD_ERR_ARG, "logp at the starting point of chain %d is not finite", chains[k]);
    }
    s->ch[(size_t)k].init(theta0 + (size_t)k * ABD_N_THETA, s->lp[(size_t)k], s->gr.data() + (size_t)k * ABD_N_THETA,
                          opts->seed, (uint64_t)((int64_t)chains[k] + opts->chain_offset), opts->tune, opts->max_treedepth, opts->target_accept,
                          opts->dense_metric != 0);
  }
  if (opts->accumulate) {
    const size_t bytes = (size_t)n * 3 * c->G * c->N * sizeof(double);
    hipError_t e = hipMalloc(&s->d_sums, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_sums, 0, bytes, c->stream);
    if (e != hipSuccess) {
      if (s->d_sums) (void)hipFree(s->d_sums);
      train_free(s);
      dtrain_free(s);
      delete s;
      return fail(ABD_ERR_HIP, "sampler sums: %s", hipGetErrorString(e));
    }
  }
  *out = s;
  return ABD_OK;
}

void abd_sampler_destroy(abd_sampler* s) {
  if (!s) return;
  if (!s->tu.empty() || !s->dc.empty()) {
    (void)hipSetDevice(s->c->device);
    (void)hipDeviceSynchronize();  // launches of a half that ended early may still be on their way
    train_free(s);
    dtrain_free(s);
  }
  if (s->d_sums || s->d_rec_mu || s->d_rec_i8) {
    (void)hipSetDevice(s->c->device);
    (void)hipStreamSynchronize(s->c->stream);
    if (s->d_sums) (void)hipFree(s->d_sums);
    if (s->d_rec_mu) (void)hipFree(s->d_rec_mu);
    if (s->d_rec_i8) (void)hipFree(s->d_rec_i8);
  }
  delete s;
}

int abd_sampler_run(abd_sampler* s, int64_t n_iter, double* theta, double* stats) {
  return abd_sampler_run_record(s, n_iter, theta, stats, nullptr);
}

namespace {

// copy staged draws [0, filled) of chain k to the caller's arrays, starting at draw `first` (stream st, waited for)
int record_flush_chain(abd_sampler* s, const abd_record* rec, int k, int64_t first, int64_t filled, hipStream_t st) {
  if (filled == 0) return ABD_OK;
  abd_ctx* c = s->c;
  const size_t cells = (size_t)c->G * c->N, N = (size_t)c->N;
  const size_t per_var = (size_t)s->n * s->rec_chunk * cells;
  const size_t dev = (size_t)k * s->rec_chunk, host = (size_t)k * rec->capacity + first;
  if (rec->ab_n_mu) HIP_TRY(hipMemcpyAsync(rec->ab_n_mu + host * cells, s->d_rec_mu + dev * cells, filled * cells * sizeof(double), hipMemcpyDeviceToHost, st));
  if (rec->ab_s_mu) HIP_TRY(hipMemcpyAsync(rec->ab_s_mu + host * cells, s->d_rec_mu + per_var + dev * cells, filled * cells * sizeof(double), hipMemcpyDeviceToHost, st));
  if (rec->i_raw) HIP_TRY(hipMemcpyAsync(rec->i_raw + host * cells, s->d_rec_i8 + dev * cells, filled * cells, hipMemcpyDeviceToHost, st));
  if (rec->i) HIP_TRY(hipMemcpyAsync(rec->i + host * cells, s->d_rec_i8 + per_var + dev * cells, filled * cells, hipMemcpyDeviceToHost, st));
  if (rec->ab_s_waner) HIP_TRY(hipMemcpyAsync(rec->ab_s_waner + host * N, s->d_rec_i8 + 2 * per_var + dev * N, filled * N, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ABD_OK;
}

// stage the current draw of chain k at position `pos` of its chunk (all asynchronous on stream st)
int record_stage_chain(abd_sampler* s, const abd_record* rec, int k, int64_t pos, hipStream_t st, double* sums = nullptr) {
  abd_ctx* c = s->c;
  const size_t cells = (size_t)c->G * c->N, N = (size_t)c->N;
  const size_t per_var = (size_t)s->n * s->rec_chunk * cells;
  const size_t at = ((size_t)k * s->rec_chunk + pos);
  const int chain = s->chains[(size_t)k];
  const ChainSlot& slot = c->slots[(size_t)chain];
  if (rec->i || rec->ab_n_mu || rec->ab_s_mu || sums)  // (sums: the draw also goes into the device-resident running sums)
    if (int rc = launch_deterministics(c, chain, s->ch[(size_t)k].nuts.q, st, rec->i ? s->d_rec_i8 + per_var + at * cells : (int8_t*)nullptr,
                                       rec->ab_n_mu ? s->d_rec_mu + at * cells : (double*)nullptr,
                                       rec->ab_s_mu ? s->d_rec_mu + per_var + at * cells : (double*)nullptr, sums))
      return rc;
  if (rec->i_raw)
    if (int rc = launch_unpack(c, chain, s->d_rec_i8 + at * cells, st)) return rc;
  if (rec->ab_s_waner)
    HIP_TRY(hipMemcpyAsync(s->d_rec_i8 + 2 * per_var + at * N, slot.waner, N, hipMemcpyDeviceToDevice, st));
  return ABD_OK;
}

}  // namespace

namespace {

// The sampler's chains run as independent UNITS of `unit` consecutive chains (1 for large dense cohorts, 4 otherwise;
// abd_sampler_create), unit u on HIP stream u mod 8 with its own private result rows (slot kSyncSlot + u):
//   tree:      one evaluation launch per leapfrog for the unit's chains whose tree is still growing
//   sweep:     when all its trees have stopped, the unit's Gibbs sweep and the evaluation at the new state are queued
//              back to back on its stream (stream order: no host wait in between), counts copied to pinned memory
//   recording: queued on the same stream behind them
// The host polls the completion tags of whatever is in flight and moves each unit's state machine on.  No unit waits
// for another one's trees -- in lock step an iteration lasts as long as the LONGEST tree of all chains -- and the
// units' launches overlap on the device.  Within a unit the chains share launches (small cohorts are launch-bound:
// ~6 us of host time per evaluation launch).  Same compound step per chain, same random streams, and the numbers a
// unit's launch produces depend only on the unit (fixed grid), never on the other units or on timing.
int sampler_run_units(abd_sampler* s, int64_t n_iter, double* theta, double* stats, const abd_record* rec, bool recording) {
  abd_ctx* c = s->c;
  const int n = s->n, B = s->unit;
  const int n_units = (n + B - 1) / B;
  const int64_t thin = recording ? std::max<int64_t>(1, rec->thin) : 1;  // iterations 0, thin, 2 thin, ... of the call are recorded
  enum { EVAL, POST, DONE };
  struct Unit {
    int lo = 0, hi = 0, m = 0, state = EVAL;
    int64_t k = 0;       // iterations completed in this call
    int64_t staged = 0;  // draws staged on the device, not yet copied out
    int64_t flushed_to = 0;
    double tag = 0.0;
    std::chrono::steady_clock::time_point t_queued;  // profile: when its last evaluation had been queued
    std::vector<int32_t> ids, who;
    std::vector<double> th, lp, gr;
  };
  std::vector<Unit> units((size_t)n_units);
  const std::chrono::steady_clock::time_point t_run_begin = std::chrono::steady_clock::now();
  // host threads (see below): a power of two <= 8, so that units that share a HIP stream (u and u + 8) share their thread
  // (one thread while abd_kernel_timing is on: the event bookkeeping of enqueue_group belongs to the context, not to a unit)
  int T_all = 1;
  while (c->timing == 0 && 2 * T_all <= std::min({s->threads, n_units, (int)kMaxPipes})) T_all *= 2;
  // one completion-tag sequence per unit, disjoint from the context's and from each other's (unit u: (u + 1) 2^40 + k).  It
  // belongs to the CONTEXT, like the result rows kSyncSlot + u the tags are compared against: monotone for the life of
  // those rows, whichever sampler drives them
  while (c->unit_seq.size() < (size_t)n_units) c->unit_seq.push_back((double)(c->unit_seq.size() + 1) * 1099511627776.0);
  s->unit_tags = T_all > 1;
  HIP_TRY(hipSetDevice(c->device));
  if (int frc = flush_ring(c)) return frc;
  HIP_TRY(hipStreamSynchronize(c->stream));  // whatever the caller queued on the context's stream comes first
  auto stream_of = [&](int u) { return c->pipe[unit_pipe(c, u)].st; };
  // evaluate the points th[0 .. m) of the unit's chains who[0 .. m)
  auto launch_eval = [&](int u) -> int {
    Unit& un = units[(size_t)u];
    if (s->trains) {  // a plain evaluation (no leapfrog: ve = 0), assembled on the device like every other of this sampler
      un.t_queued = std::chrono::steady_clock::now();
      return train_launch(s, u, un.th.data(), nullptr, 0.0, s->ch[(size_t)u].nuts.inv_mass, true);
    }
    double* seqp = T_all > 1 ? &c->unit_seq[(size_t)u] : nullptr;
    int rc = enqueue_slot(c, kSyncSlot + u, un.m, un.ids.data(), un.th.data(), true, false, unit_pipe(c, u), seqp);
    if (rc) return rc;
    un.tag = seqp ? *seqp : c->seq;
    un.t_queued = std::chrono::steady_clock::now();
    return ABD_OK;
  };
  auto launch_tree = [&](int u) -> int {  // the next leapfrog of every tree of the unit that is still growing
    Unit& un = units[(size_t)u];
    if (s->trains) {
      abdnuts::Nuts& nu = s->ch[(size_t)u].nuts;
      abd_sampler::TrainUnit& t = s->tu[(size_t)u];
      un.m = nu.active ? 1 : 0;
      if (!nu.active) {
        t.cons = t.prod;  // what is still queued for a half that ended early is never read
        return ABD_OK;
      }
      un.who[0] = u;
      un.t_queued = std::chrono::steady_clock::now();
      if (nu.n_leaf == 0) return train_begin(s, u);  // a new half: its first point is staged on the host
      if (t.queued_in_half < nu.n_target) {          // the half goes on: keep the look-ahead full
        t.queued_in_half += 1;
        return train_launch(s, u, nullptr, nullptr, nu.signed_step(), nu.inv_mass, s->lookahead == 0 || t.queued_in_half == nu.n_target);
      }
      return ABD_OK;
    }
    un.m = 0;
    for (int j = un.lo; j < un.hi; ++j) {
      abdnuts::Nuts& nu = s->ch[(size_t)j].nuts;
      if (!nu.active) continue;
      un.ids[(size_t)un.m] = s->chains[(size_t)j];
      un.who[(size_t)un.m] = j;
      std::memcpy(un.th.data() + (size_t)un.m * ABD_N_THETA, nu.request(), sizeof(double) * ABD_N_THETA);
      ++un.m;
    }
    if (!un.m) return ABD_OK;
    return launch_eval(u);
  };
  auto ready = [&](int u) -> bool {  // have all result rows of the unit's launch landed? (never blocks)
    const Unit& un = units[(size_t)u];
    if (s->trains) return train_ready(s, u);
    volatile const double* rows = c->h_out + (size_t)(kSyncSlot + u) * c->n_slots * ABD_NOUT;
    for (int k = un.m - 1; k >= 0; --k)
      if (rows[(size_t)k * ABD_NOUT + ABD_NOUT - 1] != un.tag) return false;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return true;
  };
  // end of iteration un.k of the unit's chains (points and discrete states are final): outputs, running sums, recording;
  // then the next iteration's first leapfrogs, or DONE
  auto finish_iteration = [&](int u, bool with_counts) -> int {
    Unit& un = units[(size_t)u];
    hipStream_t st = stream_of(u);
    const bool draw = s->it + un.k >= s->o.tune;
    for (int j = un.lo; j < un.hi; ++j) {
      const abdnuts::Nuts& nu = s->ch[(size_t)j].nuts;
      if (theta) std::memcpy(theta + ((size_t)j * n_iter + un.k) * ABD_N_THETA, nu.q, sizeof(double) * ABD_N_THETA);
      if (stats) {
        double* o = stats + ((size_t)j * n_iter + un.k) * ABD_N_STATS;
        o[ABD_STAT_LP] = nu.lp;
        o[ABD_STAT_TREE_DEPTH] = nu.stats.tree_depth;
        o[ABD_STAT_N_STEPS] = nu.stats.n_steps;
        o[ABD_STAT_MEAN_TREE_ACCEPT] = nu.stats.mean_tree_accept;
        o[ABD_STAT_STEP_SIZE] = nu.stats.step_size;
        o[ABD_STAT_DIVERGING] = nu.stats.diverging ? 1.0 : 0.0;
        o[ABD_STAT_ENERGY] = nu.stats.energy;
        o[ABD_STAT_MAX_ENERGY_ERROR] = nu.stats.max_energy_error;
        o[ABD_STAT_GIBBS_ACCEPTED] = with_counts ? (double)c->h_counts_chain[2 * (size_t)j] : 0.0;
        o[ABD_STAT_GIBBS_PROPOSED] = with_counts ? (double)c->h_counts_chain[2 * (size_t)j + 1] : 0.0;
        o[ABD_STAT_T_DONE] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run_begin).count();
      }
    }
    // the next iteration's first leapfrogs go out BEFORE this iteration's recording is queued: the recording kernels read the
    // point (by value) and the discrete state, which only the next sweep -- queued after them -- changes, and the host's time
    // for queueing them (several launches per draw) passes while the device is already at work for the chain
    const bool last = un.k + 1 == n_iter;
    if (!last) {
      for (int j = un.lo; j < un.hi; ++j) s->ch[(size_t)j].begin();
      un.state = EVAL;
      if (int rc = launch_tree(u)) return rc;
    }
    for (int j = un.lo; j < un.hi; ++j) {
      // running sums and the draw's record share one launch of the Deterministics kernel where both are wanted
      double* sums = (draw && s->d_sums) ? s->d_sums + (size_t)j * 3 * (size_t)c->G * c->N : nullptr;
      if (recording && un.k % thin == 0) {
        if (int rc = record_stage_chain(s, rec, j, un.staged, st, sums)) return rc;
      } else if (sums) {
        if (int rc = accumulate_chain(s, j, st)) return rc;
      }
    }
    if (recording && un.k % thin == 0 && ++un.staged == s->rec_chunk) {
      for (int j = un.lo; j < un.hi; ++j)
        if (int rc = record_flush_chain(s, rec, j, un.flushed_to, un.staged, st)) return rc;
      un.flushed_to += un.staged;
      un.staged = 0;
    }
    un.k += 1;
    if (last) {
      un.state = DONE;
      if (recording)
        for (int j = un.lo; j < un.hi; ++j)
          if (int rc = record_flush_chain(s, rec, j, un.flushed_to, un.staged, st)) return rc;
    }
    return ABD_OK;
  };
  for (int u = 0; u < n_units; ++u) {
    Unit& un = units[(size_t)u];
    un.lo = u * B;
    un.hi = std::min(n, un.lo + B);
    const size_t cap = (size_t)(un.hi - un.lo);
    un.ids.resize(cap);
    un.who.resize(cap);
    un.th.resize(cap * ABD_N_THETA);
    un.lp.resize(cap);
    un.gr.resize(cap * ABD_N_THETA);
    un.flushed_to = recording ? rec->first : 0;
    if (n_iter == 0) {
      un.state = DONE;
      continue;
    }
    for (int j = un.lo; j < un.hi; ++j) s->ch[(size_t)j].begin();
    if (int rc = launch_tree(u)) return rc;
  }
  // The units are driven by T host threads, thread t the units u = t (mod T): what a thread touches is private to its
  // units (stream, result rows, tag sequence, chains, the caller's arrays per chain) or read-only, so the threads
  // share nothing but the HIP runtime.  T = 1 for dense cohorts (the device bounds them), up to 4 for observation lists,
  // where the host's two launches per evaluation (~7 us) are what bounds a single thread.
  // ABD_SAMPLER_PROFILE=1: how much of the wall time a host thread spends handling results and queueing launches.
  static const bool profile = env_int("ABD_SAMPLER_PROFILE", 0) != 0;
  const int T = T_all;
  g_launch_profile = LaunchProfile();
  g_launch_profile.on = profile && T == 1;
  using clk = std::chrono::steady_clock;
  std::atomic<int> first_error{ABD_OK};
  std::vector<std::string> errors((size_t)T);
  auto worker = [&](int tid) -> int {
    if (hipSetDevice(c->device) != hipSuccess) return fail(ABD_ERR_HIP, "hipSetDevice failed");
    const clk::time_point t_begin = clk::now();
    clk::time_point t_handle;
    double busy_s = 0.0, prof_fetch = 0.0, prof_feed = 0.0, prof_launch = 0.0, prof_wait = 0.0;
    long handled = 0;
    for (long spins = 0;;) {
      bool any = false, progressed = false;
      if (first_error.load(std::memory_order_relaxed) != ABD_OK) return ABD_OK;  // another thread failed: stop queueing
      for (int u = tid; u < n_units; u += T) {
        Unit& un = units[(size_t)u];
        if (un.state == DONE) continue;
        any = true;
        if (progressed && profile) {  // close the previous unit's handling interval
          busy_s += std::chrono::duration<double>(clk::now() - t_handle).count();
          t_handle = clk::now();
        }
        if (!ready(u)) continue;
        if (profile && !progressed) t_handle = clk::now();
        if (profile) prof_wait += std::chrono::duration<double>(clk::now() - un.t_queued).count();
        progressed = true;
        ++handled;
        clk::time_point tp0;
        if (profile) tp0 = clk::now();
        const double *next_q = nullptr, *next_p_half = nullptr;
        if (s->trains) {  // the launch assembled its own result: take the record
          abd_sampler::TrainUnit& t = s->tu[(size_t)u];
          const TrainRecord& r = t.rec_h[t.cons % abd_sampler::kTrainRing];
          un.lp[0] = r.lp;
          std::memcpy(un.gr.data(), r.g, sizeof(double) * ABD_N_THETA);
          next_q = r.next_theta;  // (the slot is not written again before kTrainRing more launches have been queued)
          next_p_half = r.next_p_half;
          t.cons += 1;
        } else if (int frc = fetch_slot(c, kSyncSlot + u, un.lp.data(), un.gr.data())) {
          return frc;
        }
        if (profile) {
          const clk::time_point t1 = clk::now();
          prof_fetch += std::chrono::duration<double>(t1 - tp0).count();
          tp0 = t1;
        }
        if (un.state == EVAL) {
          for (int q = 0; q < un.m; ++q)
            s->ch[(size_t)un.who[(size_t)q]].nuts.feed(un.lp[(size_t)q], un.gr.data() + (size_t)q * ABD_N_THETA, next_q, next_p_half);
          if (profile) {
            const clk::time_point t1 = clk::now();
            prof_feed += std::chrono::duration<double>(t1 - tp0).count();
            tp0 = t1;
          }
          const int lrc = launch_tree(u);
          if (profile) prof_launch += std::chrono::duration<double>(clk::now() - tp0).count();
          if (lrc) return lrc;
          if (un.m) continue;  // some tree of the unit is still growing
          for (int j = un.lo; j < un.hi; ++j) s->ch[(size_t)j].end_transition();
          if (!s->o.gibbs) {
            if (int rc = finish_iteration(u, false)) return rc;
            continue;
          }
          // binary Gibbs-Metropolis on [i_raw, ab_s_waner] of the unit's chains, then logp and gradient at the new
          // states: queued back to back on the unit's stream
          hipStream_t st = stream_of(u);
          un.m = un.hi - un.lo;
          for (int j = un.lo; j < un.hi; ++j) {
            un.ids[(size_t)(j - un.lo)] = s->chains[(size_t)j];
            un.who[(size_t)(j - un.lo)] = j;
            std::memcpy(un.th.data() + (size_t)(j - un.lo) * ABD_N_THETA, s->ch[(size_t)j].nuts.q, sizeof(double) * ABD_N_THETA);
          }
          if (int rc = enqueue_gibbs(c, un.m, un.ids.data(), un.th.data(), (s->o.seed << 20) ^ 0x5EEDull, (uint32_t)(s->it + un.k),
                                     (uint32_t)s->o.chain_offset, st, c->d_counts_chain + 2 * (size_t)un.lo,
                                     c->d_work + c->n_slots + un.lo, nullptr))
            return rc;
          HIP_TRY(hipMemcpyAsync(c->h_counts_chain + 2 * (size_t)un.lo, c->d_counts_chain + 2 * (size_t)un.lo,
                                 (size_t)un.m * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
          un.state = POST;
          if (int rc = launch_eval(u)) return rc;
        } else {  // POST: the sweep and the evaluation behind it are done (the counts landed before: same stream)
          for (int q = 0; q < un.m; ++q)
            s->ch[(size_t)un.who[(size_t)q]].nuts.set_point(un.lp[(size_t)q], un.gr.data() + (size_t)q * ABD_N_THETA);
          if (int rc = finish_iteration(u, true)) return rc;
        }
      }
      if (progressed && profile) busy_s += std::chrono::duration<double>(clk::now() - t_handle).count();
      if (!any) break;
      if (progressed) {
        spins = 0;
      } else if (++spins > 4000000) {
        __atomic_fetch_add(&c->wait_fallbacks, (int64_t)1, __ATOMIC_RELAXED);  // no tag for tens of ms: synchronise the streams in flight (see abd_wait_fallbacks)
        for (int u = tid; u < n_units; u += T)
          if (units[(size_t)u].state != DONE) HIP_TRY(hipStreamSynchronize(stream_of(u)));
        spins = 0;
      } else {
        __builtin_ia32_pause();
      }
    }
    if (profile) {
      const double wall = std::chrono::duration<double>(clk::now() - t_begin).count();
      std::fprintf(stderr, "abd sampler: thread %d of %d, %d units of %d chains in all, %ld results handled in %.3f s: host busy %.0f %% "
                   "(%.2f us per result: %.2f assemble, %.2f NUTS, %.2f queueing the next evaluation); evaluation queued -> result seen %.2f us\n",
                   tid, T, n_units, B, handled, wall, 100.0 * busy_s / wall, handled ? 1e6 * busy_s / handled : 0.0,
                   handled ? 1e6 * prof_fetch / handled : 0.0, handled ? 1e6 * prof_feed / handled : 0.0,
                   handled ? 1e6 * prof_launch / handled : 0.0, handled ? 1e6 * prof_wait / handled : 0.0);
      if (g_launch_profile.on)
        std::fprintf(stderr, "abd sampler: inside hipLaunchKernelGGL: %.2f us per evaluation launch (%ld), %.2f us per sum launch (%ld)\n",
                     g_launch_profile.evals ? 1e6 * g_launch_profile.eval_s / g_launch_profile.evals : 0.0, g_launch_profile.evals,
                     g_launch_profile.sums ? 1e6 * g_launch_profile.sum_s / g_launch_profile.sums : 0.0, g_launch_profile.sums);
    }
    return ABD_OK;
  };
  auto run_worker = [&](int tid) {
    const int rc = worker(tid);
    if (rc != ABD_OK) {
      errors[(size_t)tid] = last_error();  // the message is thread-local: hand it to the calling thread
      int expected = ABD_OK;
      first_error.compare_exchange_strong(expected, rc);
    }
  };
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(run_worker, t);
    run_worker(0);
    for (auto& th : pool) th.join();
  }
  g_launch_profile.on = false;
  if (first_error.load() != ABD_OK) {
    for (int t = 0; t < T; ++t)
      if (!errors[(size_t)t].empty()) return fail(first_error.load(), "%s", errors[(size_t)t].c_str());
    return fail(first_error.load(), "sampler thread failed");
  }
  // the context's stream continues behind everything the units queued
  for (int pi = 1; pi < c->n_streams; ++pi) c->pipe[pi].busy = true;
  if (int jrc = join_pipes(c)) return jrc;
  const int64_t first_draw = std::max<int64_t>(s->it, s->o.tune);
  if (s->d_sums && s->it + n_iter > first_draw) s->n_accumulated += s->it + n_iter - first_draw;
  s->it += n_iter;
  return ABD_OK;
}

// ---- dense cohorts: the compound step over leapfrog-train units (abd_sampler::dtrains) ----
// One host thread.  Per unit (1, 2 or 4 consecutive chains, stream unit_pipe(u)): launches are queued `dtrain_lookahead`
// steps ahead; every launch takes each chain of the unit that is inside a tree one leapfrog further (TrainChainArgs STEP),
// hands a new transition to a chain that is waiting for one (BEGIN, read by the launch's last workgroup) and leaves the
// others alone.  Per chain: the records are taken in order and fed to the NUTS state machine, which runs BEHIND the device;
// when it finds the tree ended, the steps still queued for the chain are stale (their records are never looked at) and the
// chain's iteration goes on beside the unit's launches: sweep + counts on the chain's own stream, then -- the discrete
// state has changed -- a BEGIN whose first step evaluates the start point (EVAL0) before the tree starts from it.
// Nothing a chain computes depends on the other chains or on timing: launch shape and step order are fixed per chain.
int sampler_run_trains(abd_sampler* s, int64_t n_iter, double* theta, double* stats, const abd_record* rec, bool recording) {
  abd_ctx* c = s->c;
  const int n = s->n, B = s->unit;
  const int n_units = (n + B - 1) / B;
  const int64_t thin = recording ? std::max<int64_t>(1, rec->thin) : 1;
  enum { NEED_BEGIN, TREE, SWEEP, DONE };
  struct Step {
    int64_t idx;     // record index
    uint32_t epoch;  // the transition it belongs to
    bool eval0;
  };
  struct Run {
    int state = DONE;
    int64_t k = 0;           // iterations completed in this call
    uint32_t epoch = 0;      // counts the chain's transitions: steps of an earlier one are stale
    int parity = 0;          // use_slot of the chain's next step
    int steps_queued = 0, max_steps = 0;
    bool eval_first = false, eval_only = false;
    int begin_block = 0;
    int pend_slot = -1;      // the last step left its record beside pt[pend_slot] ...
    int64_t pend_idx = 0;    // ... for the next launch's service workgroup to pass on
    std::deque<Step> fifo;
    int64_t staged = 0, flushed_to = 0;  // recording: draws staged on the device / copied out
  };
  std::vector<Run> runs((size_t)n);
  HIP_TRY(hipSetDevice(c->device));
  if (int frc = flush_ring(c)) return frc;
  HIP_TRY(hipStreamSynchronize(c->stream));  // whatever the caller queued on the context's stream comes first
  static const bool profile = env_int("ABD_SAMPLER_PROFILE", 0) != 0;
  using clk = std::chrono::steady_clock;
  g_launch_profile = LaunchProfile();
  g_launch_profile.on = profile;
  long n_launches = 0, n_records = 0, n_stale = 0;
  const clk::time_point t_begin = clk::now();

  auto stage_begin = [&](int j, bool eval_first, bool eval_only) {
    // hand chain j's next transition to the device: begin_draw() has been called (momentum and directions drawn)
    Run& r = runs[(size_t)j];
    abd_sampler::DChain& d = s->dc[(size_t)j];
    const abdnuts::Nuts& nu = s->ch[(size_t)j].nuts;
    r.begin_block = (int)(d.n_begin++ % abd_sampler::kBeginBlocks);
    TrainBegin& b = d.begin_h[r.begin_block];
    std::memcpy(b.q0, nu.q, sizeof b.q0);
    std::memcpy(b.p0, nu.p0_pending, sizeof b.p0);
    std::memcpy(b.g0, nu.g, sizeof b.g0);
    std::memcpy(b.inv_mass, nu.inv_mass, sizeof b.inv_mass);
    b.eps = nu.eps;
    b.dirs = nu.dir_bits;
    b.max_depth = eval_only ? 0 : nu.max_depth;
    b.eval_first = eval_first ? 1 : 0;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    r.eval_first = eval_first;
    r.eval_only = eval_only;
    r.state = NEED_BEGIN;
  };
  // outputs of iteration r.k of chain j (point and discrete state are final), its recording, then the next transition
  auto iteration_done = [&](int j, bool with_counts) -> int {
    Run& r = runs[(size_t)j];
    abd_sampler::DChain& d = s->dc[(size_t)j];
    const abdnuts::Nuts& nu = s->ch[(size_t)j].nuts;
    if (theta) std::memcpy(theta + ((size_t)j * n_iter + r.k) * ABD_N_THETA, nu.q, sizeof(double) * ABD_N_THETA);
    if (stats) {
      double* o = stats + ((size_t)j * n_iter + r.k) * ABD_N_STATS;
      o[ABD_STAT_LP] = nu.lp;
      o[ABD_STAT_TREE_DEPTH] = nu.stats.tree_depth;
      o[ABD_STAT_N_STEPS] = nu.stats.n_steps;
      o[ABD_STAT_MEAN_TREE_ACCEPT] = nu.stats.mean_tree_accept;
      o[ABD_STAT_STEP_SIZE] = nu.stats.step_size;
      o[ABD_STAT_DIVERGING] = nu.stats.diverging ? 1.0 : 0.0;
      o[ABD_STAT_ENERGY] = nu.stats.energy;
      o[ABD_STAT_MAX_ENERGY_ERROR] = nu.stats.max_energy_error;
      o[ABD_STAT_GIBBS_ACCEPTED] = with_counts ? (double)d.done_h[0] : 0.0;
      o[ABD_STAT_GIBBS_PROPOSED] = with_counts ? (double)d.done_h[1] : 0.0;
      o[ABD_STAT_T_DONE] = std::chrono::duration<double>(clk::now() - t_begin).count();
    }
    // running sums and the draw's record: on the chain's own stream, behind its sweep and in front of the next one (they read
    // the discrete state; the point goes by value)
    const bool draw = s->it + r.k >= s->o.tune;
    double* sums = (draw && s->d_sums) ? s->d_sums + (size_t)j * 3 * (size_t)c->G * c->N : nullptr;
    if (recording && r.k % thin == 0) {
      if (int rc = record_stage_chain(s, rec, j, r.staged, d.side, sums)) return rc;
      if (++r.staged == s->rec_chunk) {
        if (int rc = record_flush_chain(s, rec, j, r.flushed_to, r.staged, d.side)) return rc;
        r.flushed_to += r.staged;
        r.staged = 0;
      }
    } else if (sums) {
      if (int rc = accumulate_chain(s, j, d.side)) return rc;
    }
    r.k += 1;
    if (r.k == n_iter) {
      r.state = DONE;
      if (recording)
        if (int rc = record_flush_chain(s, rec, j, r.flushed_to, r.staged, d.side)) return rc;
      r.staged = 0;
    }
    return ABD_OK;
  };
  // chain j's tree has ended (the NUTS state machine holds the new point): adaptation, then the sweep or the next transition
  auto transition_end = [&](int j) -> int {
    Run& r = runs[(size_t)j];
    abd_sampler::DChain& d = s->dc[(size_t)j];
    r.epoch += 1;  // what is still queued for the chain belongs to a tree that is over
    s->ch[(size_t)j].end_transition();
    if (s->o.gibbs) {
      const int32_t id = s->chains[(size_t)j];
      if (int rc = enqueue_gibbs(c, 1, &id, s->ch[(size_t)j].nuts.q, (s->o.seed << 20) ^ 0x5EEDull, (uint32_t)(s->it + r.k), (uint32_t)s->o.chain_offset,
                                 d.side, c->d_counts_chain + 2 * (size_t)j, c->d_work + c->n_slots + j, nullptr))
        return rc;
      d.sweep_tag += 1.0;
      hipLaunchKernelGGL(abd_sweep_done_kernel, dim3(1), dim3(64), 0, d.side, c->d_counts_chain + 2 * (size_t)j, d.done_d,
                         reinterpret_cast<double*>(d.done_d + 2), d.sweep_tag);
      HIP_TRY(hipGetLastError());
      r.state = SWEEP;
      return ABD_OK;
    }
    if (int rc = iteration_done(j, false)) return rc;
    if (r.state != DONE) {
      s->ch[(size_t)j].begin();  // (logp and gradient at the new point are the proposal's)
      stage_begin(j, false, false);
    }
    return ABD_OK;
  };

  for (int j = 0; j < n; ++j) {
    Run& r = runs[(size_t)j];
    r.flushed_to = recording ? rec->first : 0;
    if (n_iter == 0) continue;
    s->ch[(size_t)j].begin();
    stage_begin(j, false, false);  // logp and gradient at the chain's point are known (abd_sampler_create, or the run before)
  }

  int rc_loop = ABD_OK;
  for (long spins = 0; rc_loop == ABD_OK;) {
    bool any = false, progressed = false;
    for (int u = 0; u < n_units && rc_loop == ABD_OK; ++u) {
      const int lo = u * B, hi = std::min(n, lo + B);
      // ---- take the records that have landed ----
      for (int j = lo; j < hi && rc_loop == ABD_OK; ++j) {
        Run& r = runs[(size_t)j];
        abd_sampler::DChain& d = s->dc[(size_t)j];
        abdnuts::Nuts& nu = s->ch[(size_t)j].nuts;
        while (!r.fifo.empty() && rc_loop == ABD_OK) {
          const Step e = r.fifo.front();
          if (e.epoch != r.epoch) {  // a step the device took beyond the end of its tree
            r.fifo.pop_front();
            ++n_stale;
            continue;
          }
          const TrainRecord& tr = d.ring_h[e.idx % ABD_TRAIN_RING];
          if (*(volatile const double*)&tr.tag != (double)(e.idx + 1)) break;
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          r.fifo.pop_front();
          progressed = true;
          ++n_records;
          if (e.eval0) {
            // logp and gradient at the start point under the new discrete state: the iteration the sweep closed is complete
            nu.set_point(tr.lp, tr.g);
            if ((rc_loop = iteration_done(j, true)) != ABD_OK) break;
            if (r.state == DONE) break;  // (the evaluation was all this BEGIN asked for)
            nu.begin_finish();
            nu.adopt_request(tr.next_theta, tr.next_p_half);
          } else {
            nu.feed(tr.lp, tr.g, tr.next_theta, tr.next_p_half, true);
            if (!nu.active) rc_loop = transition_end(j);
          }
        }
      }
      if (rc_loop != ABD_OK) break;
      // ---- sweeps that have finished: the next transition starts with an evaluation at the new state ----
      for (int j = lo; j < hi; ++j) {
        Run& r = runs[(size_t)j];
        if (r.state != SWEEP) continue;
        const abd_sampler::DChain& dj = s->dc[(size_t)j];
        if (*reinterpret_cast<volatile const double*>(dj.done_h + 2) != dj.sweep_tag) continue;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        progressed = true;
        const bool last = r.k + 1 == n_iter;
        if (!last) s->ch[(size_t)j].begin_draw();
        stage_begin(j, true, last);
      }
      if (rc_loop != ABD_OK) break;
      // ---- keep the unit's launches queued ahead ----
      for (;;) {
        size_t out = 0;
        bool any_begin = false, any_step = false;
        for (int j = lo; j < hi; ++j) {
          const Run& r = runs[(size_t)j];
          out = std::max(out, r.fifo.size());
          any_begin |= r.state == NEED_BEGIN;
          any_step |= r.state == TREE && r.steps_queued < r.max_steps;
        }
        if (!any_begin && !(any_step && out < (size_t)s->dtrain_lookahead)) break;
        const bool step_now = out < (size_t)s->dtrain_lookahead;
        DenseTrainArgs a;
        std::memset(&a, 0, sizeof a);
        for (int k = 0; k < ABD_TRAIN_CB; ++k) a.tc[k].fwd_slot = -1;  // (a unit may have fewer chains than its shape: those slots stay SKIP)
        for (int j = lo; j < hi; ++j) {
          Run& r = runs[(size_t)j];
          abd_sampler::DChain& d = s->dc[(size_t)j];
          const ChainSlot& slot = c->slots[(size_t)s->chains[(size_t)j]];
          TrainChainArgs& tc = a.tc[j - lo];
          tc.st = d.st;
          tc.ring = d.ring_d;
          tc.begin = d.begin_d;
          tc.iw = slot.iw;
          tc.cnt = slot.cnt;
          tc.waner = slot.waner;
          tc.action = ABD_TR_SKIP;
          tc.fwd_slot = -1;
          if (r.state == NEED_BEGIN) {
            tc.action = ABD_TR_BEGIN;
            tc.begin = d.begin_d + r.begin_block;
            r.state = TREE;
            r.parity = 0;
            r.steps_queued = 0;
            r.max_steps = (r.eval_first ? 1 : 0) + (r.eval_only ? 0 : s->ch[(size_t)j].nuts.max_leaves());
            r.pend_slot = -1;  // (a record left there belongs to steps beyond the end of the last tree)
          } else if (r.state == TREE && r.steps_queued < r.max_steps && step_now) {
            tc.action = ABD_TR_STEP;
            tc.use_slot = r.parity;
            tc.rec_idx = d.n_rec++;
            tc.own = r.steps_queued + 1 == r.max_steps ? 1 : 0;  // nothing can follow the tree's last possible leaf
            if (r.pend_slot >= 0) {
              tc.fwd_slot = r.pend_slot;
              tc.fwd_idx = r.pend_idx;
            }
            r.pend_slot = tc.own ? -1 : (r.parity ^ 1);
            r.pend_idx = tc.rec_idx;
            r.fifo.push_back(Step{tc.rec_idx, r.epoch, r.eval_first && r.steps_queued == 0});
            r.parity ^= 1;
            r.steps_queued += 1;
          } else if (r.pend_slot >= 0) {
            tc.fwd_slot = r.pend_slot;  // the chain sits this launch out; its last record still goes to the host
            tc.fwd_idx = r.pend_idx;
            r.pend_slot = -1;
          }
        }
        if ((rc_loop = enqueue_dense_train(c, unit_pipe(c, u), B, s->dtrain_blocks, &a)) != ABD_OK) break;
        ++n_launches;
        progressed = true;
      }
      for (int j = lo; j < hi; ++j) any |= runs[(size_t)j].state != DONE;
    }
    if (rc_loop != ABD_OK || !any) break;
    if (progressed) {
      spins = 0;
    } else if (++spins > 40000000) {
      // nothing has moved for seconds: a record that never got its tag
      __atomic_fetch_add(&c->wait_fallbacks, (int64_t)1, __ATOMIC_RELAXED);
      (void)hipDeviceSynchronize();
      rc_loop = fail(ABD_ERR_STATE, "the native sampler's leapfrog trains stalled: a record never received its tag");
    } else {
      __builtin_ia32_pause();
    }
  }
  // what the device still has queued beyond the ends of the last trees must be through before anybody reuses the chains' state
  for (int u = 0; u < n_units; ++u) (void)hipStreamSynchronize(c->pipe[unit_pipe(c, u)].st);
  for (int j = 0; j < n; ++j) (void)hipStreamSynchronize(s->dc[(size_t)j].side);
  g_launch_profile.on = false;
  if (rc_loop != ABD_OK) return rc_loop;
  if (profile) {
    const double wall = std::chrono::duration<double>(clk::now() - t_begin).count();
    std::fprintf(stderr, "abd sampler (trains): %d units of %d chains, %ld launches, %ld records (%ld stale steps) in %.3f s; %.2f us inside the "
                 "launch call per launch\n", n_units, B, n_launches, n_records, n_stale, wall,
                 g_launch_profile.evals ? 1e6 * g_launch_profile.eval_s / g_launch_profile.evals : 0.0);
  }
  for (int pi = 1; pi < c->n_streams; ++pi) c->pipe[pi].busy = true;
  if (int jrc = join_pipes(c)) return jrc;
  const int64_t first_draw = std::max<int64_t>(s->it, s->o.tune);
  if (s->d_sums && s->it + n_iter > first_draw) s->n_accumulated += s->it + n_iter - first_draw;
  s->it += n_iter;
  return ABD_OK;
}

}  // namespace

int abd_sampler_run_record(abd_sampler* s, int64_t n_iter, double* theta, double* stats, const abd_record* rec) {
  if (!s) return fail(ABD_ERR_ARG, "sampler is NULL");
  if (n_iter < 0) return fail(ABD_ERR_ARG, "n_iter=%lld is negative", (long long)n_iter);
  abd_ctx* c = s->c;
  const int n = s->n;
  const bool recording = rec && (rec->i_raw || rec->ab_s_waner || rec->i || rec->ab_n_mu || rec->ab_s_mu);
  if (recording) {
    if (rec->thin < 0) return fail(ABD_ERR_ARG, "record: thin=%lld is negative", (long long)rec->thin);
    const int64_t thin = std::max<int64_t>(1, rec->thin), n_rec = (n_iter + thin - 1) / thin;  // iterations 0, thin, 2 thin, ... of the call
    if (rec->first < 0 || rec->first + n_rec > rec->capacity)
      return fail(ABD_ERR_ARG, "record: draws [%lld, %lld) do not fit capacity %lld", (long long)rec->first,
                  (long long)(rec->first + n_rec), (long long)rec->capacity);
    if (!s->d_rec_mu) {
      HIP_TRY(hipSetDevice(c->device));
      const size_t cells = (size_t)c->G * c->N;
      const size_t per_draw = (size_t)n * (cells * 18 + c->N);  // bytes staged per draw, all chains
      s->rec_chunk = std::max<int64_t>(1, std::min<int64_t>(256, (int64_t)(((size_t)256 << 20) / per_draw)));
      double* mu = nullptr;
      int8_t* i8 = nullptr;
      hipError_t e = hipMalloc(&mu, (size_t)2 * n * s->rec_chunk * cells * sizeof(double));
      if (e == hipSuccess) e = hipMalloc(&i8, (size_t)2 * n * s->rec_chunk * cells + (size_t)n * s->rec_chunk * c->N);
      if (e != hipSuccess) {
        if (mu) (void)hipFree(mu);
        return fail(ABD_ERR_HIP, "record staging: %s", hipGetErrorString(e));
      }
      s->d_rec_mu = mu;
      s->d_rec_i8 = i8;
    }
  }
  if (s->dtrains) return sampler_run_trains(s, n_iter, theta, stats, rec, recording);
  return sampler_run_units(s, n_iter, theta, stats, rec, recording);
}

int abd_sampler_means(abd_sampler* s, int32_t k, double* i_mean, double* mu_n_mean, double* mu_s_mean, int64_t* n_draws) {
  if (!s) return fail(ABD_ERR_ARG, "sampler is NULL");
  if (k < 0 || k >= s->n) return fail(ABD_ERR_ARG, "k=%d outside [0, %d)", k, s->n);
  if (!s->d_sums) return fail(ABD_ERR_STATE, "the sampler was created without accumulate");
  abd_ctx* c = s->c;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t cells = (size_t)c->G * c->N;
  double* outs[3] = {i_mean, mu_n_mean, mu_s_mean};
  const double inv = s->n_accumulated ? 1.0 / (double)s->n_accumulated : 0.0;
  for (int v = 0; v < 3; ++v) {
    if (!outs[v]) continue;
    HIP_TRY(hipMemcpy(outs[v], s->d_sums + ((size_t)k * 3 + v) * cells, cells * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < cells; ++e) outs[v][e] *= inv;
  }
  if (n_draws) *n_draws = s->n_accumulated;
  return ABD_OK;
}

int abd_sampler_adaptation(abd_sampler* s, int32_t k, double* inv_mass, double* step_size, double* metric) {
  if (!s) return fail(ABD_ERR_ARG, "sampler is NULL");
  if (k < 0 || k >= s->n) return fail(ABD_ERR_ARG, "k=%d outside [0, %d)", k, s->n);
  const abdnuts::Nuts& nu = s->ch[(size_t)k].nuts;
  if (inv_mass) std::memcpy(inv_mass, nu.inv_mass, sizeof(double) * ABD_N_THETA);
  if (step_size) *step_size = nu.eps;
  if (metric)
    for (int r = 0; r < ABD_N_THETA; ++r)
      for (int c = 0; c < ABD_N_THETA; ++c)
        metric[r * ABD_N_THETA + c] = nu.dense ? nu.cov[r][c] : (r == c ? nu.inv_mass[r] : 0.0);
  return ABD_OK;
}

int abd_sampler_set_adaptation(abd_sampler* s, int32_t k, const double* inv_mass, double step_size) {
  if (!s) return fail(ABD_ERR_ARG, "sampler is NULL");
  if (k < 0 || k >= s->n) return fail(ABD_ERR_ARG, "k=%d outside [0, %d)", k, s->n);
  abdnuts::Nuts& nu = s->ch[(size_t)k].nuts;
  if (nu.dense) return fail(ABD_ERR_STATE, "chain %d runs a dense metric", k);
  if (inv_mass) {
    for (int d = 0; d < ABD_N_THETA; ++d)
      if (!(inv_mass[d] > 0.0) || !std::isfinite(inv_mass[d])) return fail(ABD_ERR_ARG, "inv_mass[%d]=%g is not a positive finite number", d, inv_mass[d]);
    std::memcpy(nu.inv_mass, inv_mass, sizeof(double) * ABD_N_THETA);
  }
  if (step_size > 0.0) {
    if (!std::isfinite(step_size)) return fail(ABD_ERR_ARG, "step_size is not finite");
    nu.eps = step_size;
  }
  return ABD_OK;
}

}  // extern "C"
