// abd_capi.hip -- host side of the C ABI declared in include/abd_hip.h.
//
// Replaces, for the joint-logp path only, what PyMC/PyTensor compile out of abdpymc.model()
// (reference abdpymc/abd.py:396-469): the closed-form prior terms + transform Jacobians are evaluated
// here on the host (17 scalars), the O(G*N) data term on the device (abd_kernels.hpp).
#include "abd_kernels.hpp"

#include <algorithm>
#include <thread>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/abd_hip.h"
#include "abd_nuts.hpp"

static_assert(ABD_MAX_BATCH == ABD_MAX_BATCH_K, "header / kernel batch size mismatch");
static_assert(ABD_MAX_GAPS == 64 * ABD_MAXT, "header / kernel gap limit mismatch");

namespace {

thread_local std::string g_err = "";

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(ABD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

constexpr int kResultSlots = 1024;
constexpr int kSyncSlot = kResultSlots;  // private rows of the synchronous calls: they never touch a caller's slot
constexpr int kMaxPipes = 8;             // HIP streams of a context
constexpr int kMinRows = 4;
constexpr double kLog2Pi = 1.8378770664093453;  // log(2 pi)

// ABD_SAMPLER_PROFILE: time the host spends inside hipLaunchKernelGGL for evaluation launches and their sums
struct LaunchProfile {
  bool on = false;
  double eval_s = 0.0, sum_s = 0.0;
  long evals = 0, sums = 0;
};
LaunchProfile g_launch_profile;

struct AntigenDev {
  int64_t K = 0;
  void* y = nullptr;       // sparse: R[K], sorted by (ind, gap)
  void* x = nullptr;       // sparse: R[K]
  uint8_t* g = nullptr;    // sparse: gap per obs
  int32_t* ptr = nullptr;  // sparse: (N+1)
  int32_t* j = nullptr;    // sparse: individual per obs
  void* yx = nullptr;      // dense: [G][N] of {od, log_dilution}
};

struct ChainSlot {
  uint64_t* rw = nullptr;   // [nt][N] packed i_raw
  int8_t* waner = nullptr;  // [N]
  bool set = false;
};

struct Transformed {
  double p, perm_n, temp_n, rho_n, init_n, perm_s, rho_s, q, tinf, tvac, init_s;
  double b_n, d_n, sig_n, b_s, d_s, sig_s;
};

// see prepare()
struct HostTerms {
  Transformed tr;
  double L0[4], L1[4];  // -softplus(-t), -softplus(t) of theta[0], [3], [6], [7]
};

struct ResultSlot {
  int n = 0;
  bool grad = true;
  std::vector<int32_t> chains;
  std::vector<double> theta;  // n x 17
  std::vector<HostTerms> host;  // n: the host-side terms of every theta, computed when the evaluation is queued
  double tag_first = 0.0;       // completion tag of the slot's first group of <= ABD_MAX_BATCH rows (group g: tag_first + g)
};

}  // namespace

struct abd_ctx {
  int device = 0;
  int G = 0, N = 0, nt = 0, n_chunks = 1;
  int storage = ABD_STORE_F64;
  bool dense = false;
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
  uint64_t chunk_mask[3][ABD_MAXT] = {};
  AntigenDev s, n;
  uint64_t* vw = nullptr;  // [nt][N]
  uint64_t* pw = nullptr;  // [nt][N]
  double* exp2_tab = nullptr;  // dense cohorts: 2^(j/1024) (abd_dense.hpp)
  // dense kernel: how a launch shape (grid.x, ranges per workgroup) cuts the (lane group, gap) plane into ranges, built on
  // first use and kept: {first lane group, first gap, rows, 0} per range
  struct RangeTable {
    int blocks = 0, nsub = 0;
    int32_t* dev = nullptr;
  };
  std::vector<RangeTable> range_tables;
  std::mutex range_mutex;  // range_table() may be reached from several sampler threads
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
  // ABD_OBS_FUSED_SUM=1: the observation-lane kernel sums its own partial rows (last workgroup in) instead of a second
  // launch.  Off by default: it halves the host's time per evaluation (8.1 -> 4.5 us) but the hand-off inside the kernel
  // (write-through rows, one atomic round trip, coherent re-read) costs 4.4 us more than the queued second launch, and
  // the best rate the native sampler reaches on the reference's cohorts does not move
  bool obs_fused = false;
  // A sampler unit's dense launch sums its own partial rows (abd_dense.hpp; ABD_DENSE_OWN_SUM=0: second launch).  Same
  // bits; the result arrives 2-2.7 us later than from the pre-queued second launch, but the host spends 3.6 instead of
  // 7.2 us per result: config 3, evaluations/s seen by NUTS 33.6 k -> 36.9 k (1 chain), 53 k -> 59 k (8), 77 k -> 88 k (16),
  // unchanged with 4
  bool dense_own_sum = true;
  unsigned int* d_fin_count = nullptr;  // [kMaxPipes][ABD_MAX_BATCH] zeroed counters of that sum
  uint32_t ind_offset = 0;  // global index of this context's first individual (Gibbs random streams)
  bool xcd_remap = true;
  int fin_rows = 2;
  double prior_const = 0.0;
  double* h_out = nullptr;     // pinned + mapped: [kResultSlots + n_sync_slots][n_slots][ABD_NOUT]
  double* d_out = nullptr;     // device view of h_out
  double* h_done = nullptr;    // pinned + mapped: completion tag of a small stream-ordered flush (abd_wait polls it)
  double* d_done = nullptr;
  double flush_tag = 0.0;      // tag of the flush in flight, 0 = none
  bool wait_poll = true;       // ABD_WAIT_POLL=0: abd_wait always synchronises the stream
  // ABD_DIRECT_OUT: stream-ordered results go straight to the mapped host rows (as synchronous ones do) instead of a
  // device ring that abd_wait copies out: no copy kernel behind the last launch, and a caller can take a slot's result
  // as soon as its tag shows (abd_logp_dlogp_many assembles the early steps while the late ones still run)
  bool direct_out = true;
  std::vector<int> pending_slots;  // slots queued stream-ordered since the last abd_wait
  unsigned long long* d_counts = nullptr;  // [n_slots][2] Gibbs accepted / proposed
  unsigned int* d_work = nullptr;          // [2][n_slots] work queue heads of abd_gibbs_dense_kernel (second half: per-chain sweeps of the sampler)
  unsigned long long* d_counts_chain = nullptr;  // [n_slots][2] counts of the sampler's per-chain sweeps ...
  unsigned long long* h_counts_chain = nullptr;  // ... and their pinned host copy
  bool gibbs_v1 = false;                   // ABD_GIBBS_V1=1: dense cohorts use the wave-per-proposal kernel too
  int g2_refill_min = ABD_G2_REFILL_MIN, g2_tail_lanes = ABD_G2_TAIL_LANES, g2_tail_age = ABD_G2_TAIL_AGE;  // scheduler knobs of abd_gibbs_dense_kernel (ABD_G2_*)
  double* d_det = nullptr;                 // staging of abd_deterministics: mu_n, mu_s (G*N doubles each), i (G*N bytes)
  double* d_ring = nullptr;    // device-memory copy of the result ring: stream-ordered launches write here ...
  int ring_lo = 0, ring_hi = 0;  // ... and abd_wait flushes slots [ring_lo, ring_hi) to h_out with one small kernel
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
  int64_t wait_fallbacks = 0;  // synchronous calls whose completion tag never showed and that fell back to a stream synchronise
  int64_t resident_launches = 0, resident_commands = 0, resident_restarts = 0;  // abd_resident_stats
  char name[256] = {0};
};

namespace {

inline double sigmoid(double t) { return 1.0 / (1.0 + std::exp(-t)); }
inline double softplus(double t) { return std::max(t, 0.0) + std::log1p(std::exp(-std::fabs(t))); }

Transformed transform(const double* t) {
  Transformed c;
  c.p = sigmoid(t[0]);
  c.perm_n = std::exp(t[1]);
  c.temp_n = std::exp(t[2]);
  c.rho_n = sigmoid(t[3]);
  c.init_n = t[4];
  c.perm_s = std::exp(t[5]);
  c.rho_s = sigmoid(t[6]);
  c.q = sigmoid(t[7]);
  c.tinf = std::exp(t[8]);
  c.tvac = std::exp(t[9]);
  c.init_s = t[10];
  c.b_n = t[11];
  c.d_n = t[12];
  c.sig_n = std::exp(t[13]);
  c.b_s = t[14];
  c.d_s = t[15];
  c.sig_s = std::exp(t[16]);
  return c;
}

// Everything transcendental that one theta needs on the host -- the backward transforms and the softplus pairs of the four
// logit-transformed variables -- computed once, when the evaluation is QUEUED (the host is ahead of the device then), so
// that fetching a result is a few dozen multiply-adds (at config 3 the fetch of a region's results was 4 % of the region).
HostTerms prepare(const double* t) {
  HostTerms h;
  h.tr = transform(t);
  const int k4[4] = {0, 3, 6, 7};
  for (int q = 0; q < 4; ++q) {
    h.L0[q] = -softplus(-t[k4[q]]);
    h.L1[q] = -softplus(t[k4[q]]);
  }
  return h;
}

// Priors + transform log-Jacobians in closed form, and their gradient.
//   p ~ Beta(1, G-1), i_raw ~ Bernoulli(p)                         abd.py:424-427
//   ab_n_perm/temp ~ Gamma, ab_n_rho ~ Beta(10,1), ab_n_init ~ N  abd.py:329-340
//   ab_s_* likewise, ab_s_waner ~ Bernoulli(p_waner)               abd.py:367-388
//   it_*_b ~ N(-1,.5), it_*_d ~ N(2,.5), it_*_sigma ~ Exp(1)       abd.py:464-467
// theta-independent part of the priors: -lnB(1, G-1) - 2 lnB(10, 1) + sum over the Gammas of
// alpha log beta - lgamma(alpha) + the Normals' -log sd - 1/2 log 2 pi (14 lgamma calls otherwise made per
// evaluation)
double prior_constant(int G) {
  double v = -(std::lgamma(1.0) + std::lgamma((double)(G - 1)) - std::lgamma((double)G));
  v += -2.0 * (std::lgamma(10.0) + std::lgamma(1.0) - std::lgamma(11.0));
  const double gmu[5] = {2.0, 1.0, 2.0, 1.0, 1.0};
  for (int q = 0; q < 5; ++q) {
    const double al = gmu[q] * gmu[q] / 0.25, be = gmu[q] / 0.25;
    v += al * std::log(be) - std::lgamma(al);
  }
  const double nsd[6] = {1.0, 1.0, 0.5, 0.5, 0.5, 0.5};
  for (int q = 0; q < 6; ++q) v += -std::log(nsd[q]) - 0.5 * kLog2Pi;
  return v;
}

double priors(const HostTerms& h, const double* t, int G, double cells, double n1, double N, double m1, double* g /*17 or null*/,
              double prior_const) {
  double lp = prior_const;
  if (g) std::fill(g, g + ABD_N_THETA, 0.0);
  auto gamma_ab = [](double mu, double sd, double& a, double& b) {
    a = mu * mu / (sd * sd);
    b = mu / (sd * sd);
  };
  {  // theta0
    const double L0 = h.L0[0], L1 = h.L1[0], p = h.tr.p;
    const double bm1 = (double)(G - 1) - 1.0;
    lp += (bm1 == 0.0 ? 0.0 : bm1 * L1) + L0 + L1 + n1 * L0 + (cells - n1) * L1;
    if (g) g[0] = (1.0 + n1) * (1.0 - p) - p * (bm1 + 1.0 + (cells - n1));
  }
  const int gk[5] = {1, 2, 5, 8, 9};
  const double gmu[5] = {2.0, 1.0, 2.0, 1.0, 1.0};
  for (int q = 0; q < 5; ++q) {
    double al, be;
    gamma_ab(gmu[q], 0.5, al, be);
    const double gx[5] = {h.tr.perm_n, h.tr.temp_n, h.tr.perm_s, h.tr.tinf, h.tr.tvac};  // exp(t[1]), [2], [5], [8], [9]
    const double x = gx[q];
    lp += al * t[gk[q]] - be * x;
    if (g) g[gk[q]] = al - be * x;
  }
  for (int k : {3, 6}) {
    const double L0 = h.L0[k == 3 ? 1 : 2], L1 = h.L1[k == 3 ? 1 : 2], r = k == 3 ? h.tr.rho_n : h.tr.rho_s;
    lp += 9.0 * L0 + L0 + L1;
    if (g) g[k] = 10.0 * (1.0 - r) - r;
  }
  {
    const double L0 = h.L0[3], L1 = h.L1[3], q = h.tr.q;
    lp += L0 + L1 + m1 * L0 + (N - m1) * L1;
    if (g) g[7] = (1.0 + m1) * (1.0 - q) - q * (1.0 + (N - m1));
  }
  const int nk[6] = {4, 10, 11, 12, 14, 15};
  const double nmu[6] = {-2.0, -2.0, -1.0, 2.0, -1.0, 2.0};
  const double nsd[6] = {1.0, 1.0, 0.5, 0.5, 0.5, 0.5};
  for (int q = 0; q < 6; ++q) {
    const double z = (t[nk[q]] - nmu[q]) / nsd[q];
    lp += -0.5 * z * z;
    if (g) g[nk[q]] = -z / nsd[q];
  }
  for (int k : {13, 16}) {
    const double x = k == 13 ? h.tr.sig_n : h.tr.sig_s;
    lp += -x + t[k];
    if (g) g[k] = -x + 1.0;
  }
  return lp;
}

// Combine the device sums of one chain with the host-side terms.  The device accumulates
//   Q2 = sum q^2, H.. = sums of h' = q s (1 - s), QS = sum q s   with q = od - d s   (abd_kernels.hpp)
// so  ll = -1/2 Q2 / sigma^2 - K (log sigma + 1/2 log 2 pi),  d ll / d a_k = -b (d / sigma^2) h'_k.
void assemble(const abd_ctx* c, const HostTerms& h, const double* t, const double* sums, double* logp, double* grad,
              bool with_priors = true) {
  const Transformed& tr = h.tr;
  const double n1 = sums[ABD_NACC], m1 = sums[ABD_NACC + 1];
  const double cells = (double)c->G * (double)c->N;
  double lp = 0.0;
  if (with_priors)
    lp = priors(h, t, c->G, cells, n1, (double)c->N, m1, grad, c->prior_const);
  else if (grad)
    std::fill(grad, grad + ABD_N_THETA, 0.0);
  const double Kn = (double)c->n.K, Ks = (double)c->s.K;
  const double is2_n = 1.0 / (tr.sig_n * tr.sig_n), is2_s = 1.0 / (tr.sig_s * tr.sig_s);
  lp += -0.5 * is2_n * sums[A_N_Q2] - Kn * (t[13] + 0.5 * kLog2Pi);
  lp += -0.5 * is2_s * sums[A_S_Q2] - Ks * (t[16] + 0.5 * kLog2Pi);
  *logp = lp;
  if (grad) {
    const double fn = -tr.b_n * tr.d_n * is2_n, fs = -tr.b_s * tr.d_s * is2_s;
    grad[1] += fn * tr.perm_n * sums[A_N_HC];
    grad[2] += fn * tr.temp_n * sums[A_N_HU];
    grad[3] += fn * tr.temp_n * tr.rho_n * (1.0 - tr.rho_n) * sums[A_N_HD];
    grad[4] += fn * sums[A_N_H];
    grad[11] += -tr.d_n * is2_n * sums[A_N_HX];
    grad[12] += is2_n * sums[A_N_QS];
    grad[13] += is2_n * sums[A_N_Q2] - Kn;
    grad[5] += fs * tr.perm_s * sums[A_S_HC];
    grad[6] += fs * tr.rho_s * (1.0 - tr.rho_s) * sums[A_S_HD];
    grad[10] += fs * sums[A_S_H];
    grad[14] += -tr.d_s * is2_s * sums[A_S_HX];
    grad[15] += is2_s * sums[A_S_QS];
    grad[16] += is2_s * sums[A_S_Q2] - Ks;
  }
}

void assemble(const abd_ctx* c, const double* t, const double* sums, double* logp, double* grad, bool with_priors = true) {
  assemble(c, prepare(t), t, sums, logp, grad, with_priors);
}

ChainPar chain_par(const abd_ctx* c, int chain, const Transformed& tr) {
  ChainPar p;
  p.perm_n = tr.perm_n;
  p.temp_n = tr.temp_n;
  p.rho_n = tr.rho_n;
  p.init_n = tr.init_n;
  p.perm_s = tr.perm_s;
  p.rho_s = tr.rho_s;
  p.init_s = tr.init_s;
  p.b_n = tr.b_n;
  p.d_n = tr.d_n;
  p.b_s = tr.b_s;
  p.d_s = tr.d_s;
  p.rw = c->slots[chain].rw;
  p.waner = c->slots[chain].waner;
  return p;
}
ChainPar chain_par(const abd_ctx* c, int chain, const double* t) { return chain_par(c, chain, transform(t)); }

void base_args(const abd_ctx* c, EvalArgs& a) {
  std::memset(&a, 0, sizeof a);
  a.y_n = c->n.y;
  a.x_n = c->n.x;
  a.y_s = c->s.y;
  a.x_s = c->s.x;
  a.g_n = c->n.g;
  a.g_s = c->s.g;
  a.ptr_n = c->n.ptr;
  a.ptr_s = c->s.ptr;
  a.j_n = c->n.j;
  a.j_s = c->s.j;
  a.K_n = (int32_t)c->n.K;
  a.K_s = (int32_t)c->s.K;
  a.ob_n = c->ob_n;
  a.ob_s = c->ob_s;
  a.ob_c = c->ob_c;
  a.yx_n = c->n.yx;
  a.yx_s = c->s.yx;
  a.vw = c->vw;
  a.pw = c->ignore_pcr ? nullptr : c->pw;
  a.exp2_tab = c->exp2_tab;
#ifdef ABD_STAMPS
  {
    static unsigned long long* stamps = nullptr;
    if (!stamps) (void)hipHostMalloc((void**)&stamps, 4096 * 16 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    a.stamps = stamps;
    if (const char* e = std::getenv("ABD_STAMPS_PTR_OUT")) {  // the probe reads the buffer through its address
      FILE* f = std::fopen(e, "w");
      if (f) {
        std::fprintf(f, "%llu\n", (unsigned long long)(uintptr_t)stamps);
        std::fclose(f);
      }
    }
  }
#endif
  a.G = c->G;
  a.N = c->N;
  a.nt = c->nt;
  a.n_chunks = c->n_chunks;
  a.n_lg = c->n_lg;
  std::memcpy(a.chunk_mask, c->chunk_mask, sizeof a.chunk_mask);
}

size_t table_lds_bytes(int G, int cpw, int red_rows, bool exp2_tab = false) {
  return std::max<size_t>(ABD_FIN_PARTS * ABD_NOUT * sizeof(double),  // finalize scratch of the fused form
                          (size_t)(cpw * 2 + 1) * (G + 1) * sizeof(double2_t) + (size_t)red_rows * ABD_NOUT * sizeof(double) +
                              (exp2_tab ? (size_t)ABD_EXP2_TAB * sizeof(double) : 0));
}

template <typename K>
hipError_t launch_k(K kernel, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(ABD_BLOCK), lds, st, a);
  return hipGetLastError();
}

template <typename R, int C>
hipError_t launch_dense_g(bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  return grad ? launch_k(abd_dense_kernel<R, C, true>, grid, lds, st, a) : launch_k(abd_dense_kernel<R, C, false>, grid, lds, st, a);
}
template <typename R>
hipError_t launch_dense(int C, bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  switch (C) {
    case 4: return launch_dense_g<R, 4>(grad, grid, lds, st, a);
    case 2: return launch_dense_g<R, 2>(grad, grid, lds, st, a);
    default: return launch_dense_g<R, 1>(grad, grid, lds, st, a);
  }
}
template <typename R, int C>
hipError_t launch_sparse_g(bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  return grad ? launch_k(abd_sparse_kernel<R, C, true>, grid, lds, st, a) : launch_k(abd_sparse_kernel<R, C, false>, grid, lds, st, a);
}
template <typename R>
hipError_t launch_sparse(int C, bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  switch (C) {  // no 4-chain form: it needs 169 VGPRs (2 waves per SIMD) and spills 245 SGPRs (pick_cpw caps this path at 2)
    case 2: return launch_sparse_g<R, 2>(grad, grid, lds, st, a);
    default: return launch_sparse_g<R, 1>(grad, grid, lds, st, a);
  }
}

template <typename R>
hipError_t launch_obs(bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  return grad ? launch_k(abd_obs_kernel<R, true>, grid, lds, st, a) : launch_k(abd_obs_kernel<R, false>, grid, lds, st, a);
}

int pick_cpw(const abd_ctx* c, int n) {
  const int forced = c->cpw_forced;
  if (forced == 1 || forced == 2 || forced == 4) {
    if (n % forced == 0) return forced;
  }
  if (n % 4 == 0) return 4;
  if (n % 2 == 0) return 2;
  return 1;
}

// grid of the dense kernel: an exact multiple of the CU count (every wave slot gets the same number of
// gap rows), capped so a slot has at least kMinRows rows
// share: 0 = the launch has the chip to itself, 1 = it is one of n_pipes stream-ordered launches in flight,
// 2 = it is one of the native sampler's chain groups in flight (c->group_blocks: the chip divided by their number)
int dense_blocks(const abd_ctx* c, int cpw, int share = 0, int grid_rows = 1) {
  const int nsub = ABD_WAVES_PER_BLOCK / cpw;
  const int64_t rows = (int64_t)c->n_lg * c->G;
  const int64_t cap = std::max<int64_t>(1, rows / ((int64_t)kMinRows * nsub));
  // a launch with several grid rows (more than 4 chains) fills the chip with fewer, longer ranges per row
  const int64_t alone = std::max<int64_t>(c->n_cu, c->dense_blocks / std::max(1, grid_rows));
  const int64_t half = std::max<int64_t>(c->n_cu, c->group_blocks / std::max(1, grid_rows));
  const int64_t want = share == 1 ? (int64_t)c->pipe_blocks : (share == 2 ? half : alone);
  return (int)std::max<int64_t>(1, std::min<int64_t>({want, cap, (int64_t)c->blocks_max}));
}

// The ranges of a dense launch of `blocks` workgroups x `nsub` ranges each: equal shares (+-1) of the n_lg x G rows; with
// one range per workgroup (nsub == 1) the first ABD_MAX_BATCH ranges -- the workgroups that may carry the fused
// fixed-order sum of an earlier launch -- are fin_rows shorter and the others share the difference.
int range_table(abd_ctx* c, int blocks, int nsub, const int32_t** out) {
  std::lock_guard<std::mutex> lock(c->range_mutex);
  for (const auto& rt : c->range_tables)
    if (rt.blocks == blocks && rt.nsub == nsub) {
      *out = rt.dev;
      return ABD_OK;
    }
  const int64_t rows_total = (int64_t)c->n_lg * c->G, n_ranges = (int64_t)blocks * nsub;
  const int64_t n_short = nsub == 1 ? std::min<int64_t>(ABD_MAX_BATCH, n_ranges) : 0;
  const int64_t e_fin = (nsub == 1 && (rows_total + n_short * c->fin_rows) / n_ranges >= 2 * c->fin_rows) ? c->fin_rows : 0;
  const int64_t virt = rows_total + n_short * e_fin;
  auto start = [&](int64_t r) { return r * virt / n_ranges - e_fin * std::min(r, n_short); };
  std::vector<int32_t> tab((size_t)n_ranges * 4);
  for (int64_t r = 0; r < n_ranges; ++r) {
    const int64_t pos = start(r), end = start(r + 1);
    tab[(size_t)r * 4 + 0] = (int32_t)(pos / c->G);
    tab[(size_t)r * 4 + 1] = (int32_t)(pos % c->G);
    tab[(size_t)r * 4 + 2] = (int32_t)std::max<int64_t>(0, end - pos);
    tab[(size_t)r * 4 + 3] = 0;
  }
  abd_ctx::RangeTable rt;
  rt.blocks = blocks;
  rt.nsub = nsub;
  HIP_TRY(hipMalloc(&rt.dev, tab.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(rt.dev, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  c->range_tables.push_back(rt);
  *out = rt.dev;
  return ABD_OK;
}

// Which of the context's streams can have kernels on the device at the same time?  HIP multiplexes its streams over a few
// hardware queues (4 by default) and a queue runs one kernel after the other, whichever stream it came from.  One wave
// per stream that stays for 150 us, launched back to back: a stream whose wave starts only when an earlier stream's
// wave has ended shares that stream's queue.  ~0.5 ms, once per context.
int probe_stream_queues(abd_ctx* c) {
  if (c->n_queues > 0) return ABD_OK;
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, (size_t)kMaxPipes * 2 * sizeof(unsigned long long)));
  const int ns = c->n_streams;
  // Stream pi's wave stays for 150 + 20 pi us, so the waves end at least 20 us apart and a wave that had to wait for a
  // queue starts within a few us of exactly one earlier wave's end: it is behind that one.  A wave that starts while all
  // earlier ones are still there has a queue to itself.  A host hiccup between two launches can make a stream look
  // queued, never the other way round: up to three attempts, the one that finds the most queues counts.
  int best_nq = 0, best[kMaxPipes] = {};
  hipError_t le = hipSuccess;
  for (int attempt = 0; attempt < 3 && best_nq < 4 && le == hipSuccess; ++attempt) {
    le = hipDeviceSynchronize();
    for (int pi = 0; pi < ns && le == hipSuccess; ++pi) {
      hipLaunchKernelGGL(abd_spin_kernel, dim3(1), dim3(64), 0, c->pipe[pi].st, d + 2 * pi, 15000ull + 2000ull * (unsigned long long)pi);
      le = hipGetLastError();
    }
    if (le == hipSuccess) le = hipDeviceSynchronize();
    unsigned long long h[kMaxPipes * 2] = {};
    if (le == hipSuccess) le = hipMemcpy(h, d, sizeof(unsigned long long) * 2 * (size_t)ns, hipMemcpyDeviceToHost);
    if (le != hipSuccess) break;
    int nq = 0, q_of[kMaxPipes] = {};
    unsigned long long busy_until[kMaxPipes] = {};
    for (int j = 0; j < ns; ++j) {
      int q = -1;
      for (int k = 0; k < nq && q < 0; ++k)
        if (h[2 * j] + 300 >= busy_until[k] && h[2 * j] <= busy_until[k] + 800) q = k;  // started 0-8 us after queue k drained
      if (q < 0) q = nq++;
      q_of[j] = q;
      busy_until[q] = h[2 * j + 1];
    }
    if (nq > best_nq) {
      best_nq = nq;
      std::copy(q_of, q_of + kMaxPipes, best);
    }
  }
  (void)hipFree(d);
  HIP_TRY(le);
  const int nq = best_nq;
  std::copy(best, best + kMaxPipes, c->queue_of_pipe);
  c->n_queues = std::max(1, nq);
  // the sampler's unit u runs on stream pipe_order[u]: streams of different queues first, so that as many units as
  // there are queues really run side by side
  int k = 0;
  bool taken[kMaxPipes] = {};
  for (int round = 0; k < c->n_streams; ++round) {
    bool seen[kMaxPipes] = {};
    for (int pi = 0; pi < c->n_streams; ++pi)
      if (!taken[pi] && !seen[c->queue_of_pipe[pi]]) {
        seen[c->queue_of_pipe[pi]] = true;
        taken[pi] = true;
        c->pipe_order[k++] = pi;
      }
  }
  return ABD_OK;
}

// the HIP stream (pipe) of the native sampler's unit u
inline int unit_pipe(const abd_ctx* c, int u) { return c->pipe_order[u % c->n_streams]; }

// queue the standalone fixed-order sum of a launch whose partials are still pending
int flush_pipe(abd_ctx* c, int pi) {
  abd_ctx::Pipe& p = c->pipe[pi];
  if (p.on) {
    std::chrono::steady_clock::time_point lp0;
    if (g_launch_profile.on) lp0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(abd_finalize_kernel, dim3(p.n), dim3(ABD_FIN_THREADS), 0, p.st, p.partials[p.buf], p.blocks, p.out, p.tag);
    if (g_launch_profile.on) {
      g_launch_profile.sum_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - lp0).count();
      g_launch_profile.sums++;
    }
    HIP_TRY(hipGetLastError());
    p.on = false;
  }
  return ABD_OK;
}

// pipe 0 continues only after everything queued on pipe 1 has finished
int join_pipes(abd_ctx* c) {
  for (int pi = 1; pi < c->n_streams; ++pi) {
    abd_ctx::Pipe& p = c->pipe[pi];
    if (!p.st) continue;
    if (int rc = flush_pipe(c, pi)) return rc;
    if (p.busy) {
      HIP_TRY(hipEventRecord(c->join_ev[pi], p.st));
      HIP_TRY(hipStreamWaitEvent(c->stream, c->join_ev[pi], 0));
      p.busy = false;
    }
  }
  c->next_pipe = 0;
  return ABD_OK;
}

int flush_pending(abd_ctx* c) {
  if (int rc = flush_pipe(c, 0)) return rc;
  return join_pipes(c);
}

// Enqueue the evaluation of `n` chains (n <= ABD_MAX_BATCH); their sums go to rows d_out_rows[0..n).
int enqueue_group(abd_ctx* c, int n, const int32_t* chains, const double* theta, bool grad, double* d_out_rows,
                  bool deferred = false, int force_pipe = -1, const HostTerms* host = nullptr, double* seqp = nullptr) {
  // completion tags: the context's sequence, or the caller's own (a sampler unit handled by its own host thread: its
  // result rows are private, so its tags only have to be unique among themselves)
  double& seq = seqp ? *seqp : c->seq;
  EvalArgs a;
  base_args(c, a);
  a.n_chains = n;
  for (int k = 0; k < n; ++k)
    a.ch[k] = host ? chain_par(c, chains[k], host[k].tr) : chain_par(c, chains[k], theta + (size_t)k * ABD_N_THETA);
  const bool lanes = !c->dense && c->obs_lanes;
  int cpw = lanes ? 1 : pick_cpw(c, n);
  if (!lanes && !c->dense) cpw = std::min(cpw, 2);  // wave-per-individual list kernel: see launch_sparse
  // stream-ordered dense launches rotate over the pipes; everything else runs on pipe 0 after a join
  const bool rotate = deferred && c->n_pipes > 1 && c->dense && c->fuse_finalize && c->timing != 1;  // timing 1: one launch at a time
  int blocks;
  size_t lds;
  if (lanes) {
    blocks = c->ob_n + c->ob_s + c->ob_c;
    lds = abd_obs_lds_head(c->G) + (c->obs_fused ? (size_t)ABD_FIN_PARTS * ABD_NOUT * sizeof(double) : 0);
  } else if (c->dense) {
    blocks = dense_blocks(c, cpw, force_pipe >= 0 ? 2 : (rotate ? 1 : 0), n / cpw);
    lds = table_lds_bytes(c->G, cpw, ABD_WAVES_PER_BLOCK, true);
  } else {
    blocks = c->blocks_x;
    lds = table_lds_bytes(c->G, cpw, ABD_WAVES_PER_BLOCK * cpw);
  }
  if (blocks > c->blocks_max) return fail(ABD_ERR_STATE, "internal: grid %d exceeds partial rows %d", blocks, c->blocks_max);
  if (c->dense && !lanes)
    if (int rrc = range_table(c, blocks, ABD_WAVES_PER_BLOCK / cpw, &a.range_tab)) return rrc;
  dim3 grid(blocks, n / cpw);
  int pi = 0;
  if (force_pipe >= 0) {
    pi = force_pipe;  // the caller keeps several synchronous groups in flight, one per pipe (abd_sampler_run_record)
  } else if (rotate) {
    pi = c->pipe_order[c->next_pipe];  // streams of different hardware queues (identity until probe_stream_queues has run)
    c->next_pipe = (c->next_pipe + 1) % c->n_pipes;
  } else if (int jrc = join_pipes(c)) {
    return jrc;
  }
  abd_ctx::Pipe& pp = c->pipe[pi];
  if (pi > 0) pp.busy = true;
  const int buf = pp.pbuf;
  pp.pbuf ^= 1;
  a.partials = pp.partials[buf];
  // the kernel sums its own partial rows, no second launch: observation lanes (ABD_OBS_FUSED_SUM) / a sampler unit's dense launch (ABD_DENSE_OWN_SUM)
  const bool fused_sum = (lanes && c->obs_fused) || (c->dense && !lanes && force_pipe >= 0 && c->dense_own_sum && !(c->fuse_finalize && pp.on));
  if (fused_sum) {
    a.fin_count = c->d_fin_count + (size_t)pi * ABD_MAX_BATCH;
    a.fin_out = d_out_rows;
    a.fin_tag = seq + 1.0;
  }
  a.fin_rows = c->fin_rows;
  a.xcd_remap = c->xcd_remap ? 1 : 0;
  if (c->dense && c->fuse_finalize && pp.on && pp.n <= blocks) {
    // this launch's first workgroups sum the partials of the previous launch on the same pipe
    a.prev_partials = pp.partials[pp.buf];
    a.prev_out = pp.out;
    a.prev_n_chains = pp.n;
    a.prev_blocks = pp.blocks;
    a.prev_tag = pp.tag;
    pp.on = false;
  } else {
    int frc = flush_pipe(c, pi);
    if (frc) return frc;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->timing == 1 || (c->timing == 2 && deferred && !c->win_open)) {
    if (c->ev_used == c->ev_pool.size()) {
      hipEvent_t a0, a1;
      HIP_TRY(hipEventCreate(&a0));
      HIP_TRY(hipEventCreate(&a1));
      c->ev_pool.emplace_back(a0, a1);
    }
    e0 = c->ev_pool[c->ev_used].first;
    e1 = c->ev_pool[c->ev_used].second;
    if (c->timing == 1) c->ev_used++;
    // window mode: every pipe is idle here (the previous abd_wait joined and synchronised them), so the stream of
    // the window's first launch carries its start; the end is recorded by flush_ring once all pipes have joined
    HIP_TRY(hipEventRecord(e0, pp.st));
    if (c->timing == 2) c->win_open = true;
  }
  if (c->timing == 2 && deferred) c->win_launches++;
  hipError_t le;
  std::chrono::steady_clock::time_point lp0;
  if (g_launch_profile.on) lp0 = std::chrono::steady_clock::now();
  if (lanes)
    le = c->storage == ABD_STORE_F32 ? launch_obs<float>(grad, grid, lds, pp.st, a)
                                     : launch_obs<double>(grad, grid, lds, pp.st, a);
  else if (c->dense)
    le = c->storage == ABD_STORE_F32 ? launch_dense<float>(cpw, grad, grid, lds, pp.st, a)
                                     : launch_dense<double>(cpw, grad, grid, lds, pp.st, a);
  else
    le = c->storage == ABD_STORE_F32 ? launch_sparse<float>(cpw, grad, grid, lds, pp.st, a)
                                     : launch_sparse<double>(cpw, grad, grid, lds, pp.st, a);
  if (g_launch_profile.on) {
    g_launch_profile.eval_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - lp0).count();
    g_launch_profile.evals++;
  }
  if (c->timing == 1) HIP_TRY(hipEventRecord(e1, pp.st));
  HIP_TRY(le);
  seq += 1.0;
  if (fused_sum) return ABD_OK;
  pp.on = true;
  pp.buf = buf;
  pp.n = n;
  pp.blocks = blocks;
  pp.out = d_out_rows;
  pp.tag = seq;
  if (!(c->dense && c->fuse_finalize)) return flush_pipe(c, pi);
  return ABD_OK;
}

int flush_ring(abd_ctx* c) {
  if (c->win_open) {
    // timing 2: the window ends when the last pipe has finished its last launch and that launch's sum -- what a caller
    // that polls the completion tags waits for.  One end event per pipe, behind its pending sum and BEFORE the joins
    // (the joins' barrier packets on the context's stream come after the results and are not part of the work).
    const size_t w = c->ev_used;
    if (c->win_end.size() < (w + 1) * kMaxPipes) {
      const size_t old_n = c->win_end.size();
      c->win_end.resize((w + 1) * kMaxPipes, nullptr);
      for (size_t k = old_n; k < c->win_end.size(); ++k) HIP_TRY(hipEventCreate(&c->win_end[k]));
    }
    if (c->win_mask.size() < w + 1) c->win_mask.resize(w + 1, 0u);
    c->win_mask[w] = 0u;
    for (int pi = 0; pi < c->n_streams; ++pi) {
      if (!c->pipe[pi].st || !(pi == 0 || c->pipe[pi].busy || c->pipe[pi].on)) continue;  // no work of this window on it
      if (int prc = flush_pipe(c, pi)) return prc;
      HIP_TRY(hipEventRecord(c->win_end[w * kMaxPipes + pi], c->pipe[pi].st));
      c->win_mask[w] |= 1u << pi;
    }
    c->ev_used++;
    c->win_open = false;
  }
  if (int prc = flush_pending(c)) return prc;
  if (c->ring_lo < c->ring_hi) {
    const size_t row = (size_t)c->n_slots * ABD_NOUT;
    const int64_t count = (int64_t)(c->ring_hi - c->ring_lo) * row;
    if (c->wait_poll && count <= 16384) {
      // small flush: one workgroup copies and then raises a tag the host polls (abd_wait) -- a stream synchronise
      // returns several microseconds after the stream has drained
      c->seq += 1.0;
      c->flush_tag = c->seq;
      hipLaunchKernelGGL(abd_copy_tag_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_ring + (size_t)c->ring_lo * row,
                         c->d_out + (size_t)c->ring_lo * row, count, c->d_done, c->flush_tag);
    } else {
      const int blocks = (int)std::min<int64_t>((count + 255) / 256, 1024);
      hipLaunchKernelGGL(abd_copy_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_ring + (size_t)c->ring_lo * row,
                         c->d_out + (size_t)c->ring_lo * row, count);
    }
    HIP_TRY(hipGetLastError());
    c->ring_lo = c->ring_hi = 0;
  }
  return ABD_OK;
}

// Wait for the rows of a synchronous call (written into mapped host memory) by polling their completion tag;
// falls back to a stream synchronise if it does not show up quickly.
int wait_rows(abd_ctx* c, int slot, int n, double tag, hipStream_t st = nullptr) {
  volatile const double* rows = c->h_out + (size_t)slot * c->n_slots * ABD_NOUT;
  // every row is written by its own workgroup (row, system-scope fence, tag), in no particular order: wait for
  // each tag.  Rows of an earlier group of the same call carry a smaller tag and count as landed once a later
  // group's rows are there (groups complete in stream order), so only the last group's tag value is awaited.
  const int first = ((n - 1) / ABD_MAX_BATCH) * ABD_MAX_BATCH;
  int k = n - 1;
  for (int spin = 0; spin < 2000000; ++spin) {
    while (k >= first && rows[(size_t)k * ABD_NOUT + ABD_NOUT - 1] == tag) --k;
    if (k < first) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return ABD_OK;
    }
    __builtin_ia32_pause();
  }
  // the tag did not show within ~2 M polls (tens of ms): not an error -- the stream synchronise below is always
  // correct -- but it should never happen, so it is counted (abd_wait_fallbacks) instead of passing as a slow call
  c->wait_fallbacks++;
  HIP_TRY(hipStreamSynchronize(st ? st : c->stream));
  return ABD_OK;
}

// Wait until every row of a stream-ordered slot carries its completion tag (direct_out: the rows are in mapped host memory).
// Polls; after a second hands over to a synchronise of the context's stream (which every pipe has joined by then).
int wait_slot(abd_ctx* c, int slot) {
  const ResultSlot& r = c->results[slot];
  volatile const double* rows = c->h_out + (size_t)slot * c->n_slots * ABD_NOUT;
  const auto t_poll = std::chrono::steady_clock::now();
  int k = r.n - 1;
  for (long spin = 0;; ++spin) {
    while (k >= 0 && rows[(size_t)k * ABD_NOUT + ABD_NOUT - 1] == r.tag_first + (double)(k / ABD_MAX_BATCH)) --k;
    if (k < 0) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return ABD_OK;
    }
    __builtin_ia32_pause();
    if ((spin & 4095) == 4095 && std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(1)) break;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ABD_OK;
}

int check_chains(abd_ctx* c, int n, const int32_t* chains) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (n < 1 || n > c->n_slots) return fail(ABD_ERR_ARG, "n=%d outside [1, n_chain_slots=%d]", n, c->n_slots);
  for (int k = 0; k < n; ++k) {
    if (chains[k] < 0 || chains[k] >= c->n_slots) return fail(ABD_ERR_ARG, "chain %d outside [0, %d)", chains[k], c->n_slots);
    if (!c->slots[chains[k]].set) return fail(ABD_ERR_STATE, "chain slot %d has no discrete state (call abd_set_discrete)", chains[k]);
  }
  return ABD_OK;
}

int enqueue_slot(abd_ctx* c, int slot, int n, const int32_t* chains, const double* theta, bool grad, bool deferred = false,
                 int force_pipe = -1, double* seqp = nullptr) {
  if (slot < 0 || slot >= kSyncSlot + c->n_sync_slots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slot, kResultSlots);
  int rc = check_chains(c, n, chains);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  ResultSlot& r = c->results[slot];
  r.n = n;
  r.grad = grad;
  r.chains.assign(chains, chains + n);
  r.theta.assign(theta, theta + (size_t)n * ABD_N_THETA);
  r.host.resize((size_t)n);
  for (int k = 0; k < n; ++k) r.host[(size_t)k] = prepare(theta + (size_t)k * ABD_N_THETA);
  // a synchronous call lets the finalize kernel write straight into mapped host memory (one PCIe write,
  // ~3 us inside the kernel); stream-ordered calls write device memory and are flushed together at abd_wait
  double* rows = ((deferred && !c->direct_out) ? c->d_ring : c->d_out) + (size_t)slot * c->n_slots * ABD_NOUT;
  r.tag_first = (seqp ? *seqp : c->seq) + 1.0;
  if (deferred && c->direct_out) c->pending_slots.push_back(slot);
  if (deferred && !c->direct_out) {
    if (c->ring_lo == c->ring_hi) {
      c->ring_lo = slot;
      c->ring_hi = slot + 1;
    } else {
      c->ring_lo = std::min(c->ring_lo, slot);
      c->ring_hi = std::max(c->ring_hi, slot + 1);
    }
  }
  for (int k0 = 0; k0 < n; k0 += ABD_MAX_BATCH) {
    const int m = std::min(ABD_MAX_BATCH, n - k0);
    rc = enqueue_group(c, m, chains + k0, theta + (size_t)k0 * ABD_N_THETA, grad, rows + (size_t)k0 * ABD_NOUT, deferred, force_pipe,
                       r.host.data() + k0, seqp);
    if (!rc && force_pipe >= 0) rc = flush_pipe(c, force_pipe);  // a group's fixed-order sum follows on its own stream
    if (rc) return rc;
  }
  return ABD_OK;
}

int fetch_slot(abd_ctx* c, int slot, double* logp, double* grad, bool with_priors = true) {
  if (slot < 0 || slot >= kSyncSlot + c->n_sync_slots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slot, kResultSlots);
  const ResultSlot& r = c->results[slot];
  if (r.n == 0) return fail(ABD_ERR_STATE, "result slot %d is empty", slot);
  const double* rows = c->h_out + (size_t)slot * c->n_slots * ABD_NOUT;
  for (int k = 0; k < r.n; ++k)
    assemble(c, r.host[(size_t)k], r.theta.data() + (size_t)k * ABD_N_THETA, rows + (size_t)k * ABD_NOUT, logp + k,
             (grad && r.grad) ? grad + (size_t)k * ABD_N_THETA : nullptr, with_priors);
  return ABD_OK;
}

// one antigen's observations sorted by (ind, gap)
struct SortedObs {
  std::vector<int64_t> order;  // order[k] = original index
  std::vector<int32_t> ptr;    // (N+1)
  bool one_per_cell = false;
};

int sort_obs(const abd_antigen_obs& o, int G, int N, const char* tag, SortedObs& out) {
  if (o.n_obs < 0) return fail(ABD_ERR_ARG, "%s: negative n_obs", tag);
  if (o.n_obs > 0 && (!o.idx_gap || !o.idx_ind || !o.log_dilution || !o.od)) return fail(ABD_ERR_ARG, "%s: NULL observation array", tag);
  if (o.n_obs >= (int64_t)std::numeric_limits<int32_t>::max()) return fail(ABD_ERR_ARG, "%s: too many observations", tag);
  const int64_t cells = (int64_t)G * N;
  std::vector<int32_t> count((size_t)cells + 1, 0);
  for (int64_t k = 0; k < o.n_obs; ++k) {
    const int64_t g = o.idx_gap[k], j = o.idx_ind[k];
    if (g < 0 || g >= G || j < 0 || j >= N) return fail(ABD_ERR_ARG, "%s: observation %lld has (gap=%lld, ind=%lld) outside (%d, %d)", tag, (long long)k, (long long)g, (long long)j, G, N);
    count[(size_t)(j * G + g) + 1]++;
  }
  out.one_per_cell = o.n_obs == cells;
  for (int64_t cidx = 0; cidx < cells; ++cidx) {
    if (count[(size_t)cidx + 1] != 1) out.one_per_cell = false;
    count[(size_t)cidx + 1] += count[(size_t)cidx];
  }
  out.order.resize((size_t)o.n_obs);
  std::vector<int32_t> cursor(count.begin(), count.end() - 1);
  for (int64_t k = 0; k < o.n_obs; ++k) {
    const int64_t cell = (int64_t)o.idx_ind[k] * G + o.idx_gap[k];
    out.order[(size_t)cursor[(size_t)cell]++] = k;  // stable
  }
  out.ptr.resize((size_t)N + 1);
  for (int j = 0; j <= N; ++j) out.ptr[(size_t)j] = count[(size_t)j * G];
  return ABD_OK;
}

template <typename R>
int upload_antigen(abd_ctx* c, const abd_antigen_obs& o, const SortedObs& so, AntigenDev& d) {
  const size_t K = (size_t)o.n_obs;
  const int G = c->G, N = c->N;
  d.K = o.n_obs;
  if (c->dense) {
    // gap-major panel of {od, log_dilution} pairs: element (g, j) at [g * N + j]
    std::vector<YX<R>> yx((size_t)G * N);
    for (int j = 0; j < N; ++j)
      for (int g = 0; g < G; ++g) {
        const int64_t src = so.order[(size_t)j * G + g];
        yx[(size_t)g * N + j] = YX<R>{(R)o.od[src], (R)o.log_dilution[src]};
      }
    HIP_TRY(hipMalloc(&d.yx, yx.size() * sizeof(YX<R>)));
    HIP_TRY(hipMemcpy(d.yx, yx.data(), yx.size() * sizeof(YX<R>), hipMemcpyHostToDevice));
    return ABD_OK;
  }
  std::vector<R> y(std::max<size_t>(K, 1)), x(std::max<size_t>(K, 1));
  std::vector<uint8_t> g(std::max<size_t>(K, 1));
  std::vector<int32_t> jj(std::max<size_t>(K, 1));
  for (size_t k = 0; k < K; ++k) {
    const int64_t src = so.order[k];
    y[k] = (R)o.od[src];
    x[k] = (R)o.log_dilution[src];
    g[k] = (uint8_t)o.idx_gap[src];
    jj[k] = (int32_t)o.idx_ind[src];
  }
  HIP_TRY(hipMalloc(&d.j, jj.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d.j, jj.data(), jj.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&d.y, y.size() * sizeof(R)));
  HIP_TRY(hipMalloc(&d.x, x.size() * sizeof(R)));
  HIP_TRY(hipMemcpy(d.y, y.data(), y.size() * sizeof(R), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.x, x.data(), x.size() * sizeof(R), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&d.g, g.size()));
  HIP_TRY(hipMemcpy(d.g, g.data(), g.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&d.ptr, so.ptr.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d.ptr, so.ptr.data(), so.ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return ABD_OK;
}

// (N, G) row-major 0/1 bytes (TiterData.vacs / .pcrpos) -> packed words [nt][N]
std::vector<uint64_t> pack_ng(const int8_t* src, int G, int N, int nt) {
  std::vector<uint64_t> w((size_t)nt * N, 0);
  if (!src) return w;
  for (int j = 0; j < N; ++j)
    for (int g = 0; g < G; ++g)
      if (src[(size_t)j * G + g]) w[(size_t)(g >> 6) * N + j] |= 1ull << (g & 63);
  return w;
}

void free_ctx(abd_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto* a : {&c->s, &c->n}) {
    if (a->y) (void)hipFree(a->y);
    if (a->x) (void)hipFree(a->x);
    if (a->g) (void)hipFree(a->g);
    if (a->ptr) (void)hipFree(a->ptr);
    if (a->j) (void)hipFree(a->j);
    if (a->yx) (void)hipFree(a->yx);
  }
  if (c->vw) (void)hipFree(c->vw);
  if (c->pw) (void)hipFree(c->pw);
  if (c->exp2_tab) (void)hipFree(c->exp2_tab);
  for (auto& rt : c->range_tables)
    if (rt.dev) (void)hipFree(rt.dev);
  if (c->stage_gn) (void)hipFree(c->stage_gn);
  for (auto& s : c->slots) {
    if (s.rw) (void)hipFree(s.rw);
    if (s.waner) (void)hipFree(s.waner);
  }
  for (int pi = 1; pi < kMaxPipes; ++pi)
    if (c->pipe[pi].st) (void)hipStreamSynchronize(c->pipe[pi].st);
  for (int pi = 0; pi < kMaxPipes; ++pi)
    for (int b = 0; b < 2; ++b)
      if (c->pipe[pi].partials[b]) (void)hipFree(c->pipe[pi].partials[b]);
  for (int pi = 1; pi < kMaxPipes; ++pi) {
    if (c->join_ev[pi]) (void)hipEventDestroy(c->join_ev[pi]);
    if (c->pipe[pi].st) (void)hipStreamDestroy(c->pipe[pi].st);
  }
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->h_done) (void)hipHostFree(c->h_done);
  if (c->d_ring) (void)hipFree(c->d_ring);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->d_work) (void)hipFree(c->d_work);
  if (c->d_counts_chain) (void)hipFree(c->d_counts_chain);
  if (c->d_fin_count) (void)hipFree(c->d_fin_count);
  if (c->h_counts_chain) (void)hipHostFree(c->h_counts_chain);
  if (c->d_det) (void)hipFree(c->d_det);
  for (auto& e : c->win_end)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev_pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // namespace

extern "C" {

const char* abd_version(void) { return "abdpymc_amd hip gfx950 0.2"; }

const char* abd_last_error(void) { return g_err.c_str(); }

int abd_create(const abd_desc* d, abd_ctx** out) {
  if (!d || !out) return fail(ABD_ERR_ARG, "desc / out is NULL");
  *out = nullptr;
  const int G = d->n_gaps, N = d->n_inds;
  if (G < 2) return fail(ABD_ERR_ARG, "n_gaps must be >= 2 (Beta(1, n_gaps - 1) prior on p), got %d", G);
  if (G > ABD_MAX_GAPS) return fail(ABD_ERR_ARG, "n_gaps=%d exceeds ABD_MAX_GAPS=%d", G, ABD_MAX_GAPS);
  if (N < 1) return fail(ABD_ERR_ARG, "n_inds must be >= 1, got %d", N);
  if ((int64_t)G * N >= (int64_t)1 << 31) return fail(ABD_ERR_ARG, "n_gaps*n_inds too large");
  if (d->n_chain_slots < 1) return fail(ABD_ERR_ARG, "n_chain_slots must be >= 1");
  if (d->storage != ABD_STORE_F64 && d->storage != ABD_STORE_F32) return fail(ABD_ERR_ARG, "unknown storage %d", d->storage);
  if (!d->vacs) return fail(ABD_ERR_ARG, "vacs is NULL");
  // check_splits (abd.py:604-622) -- same conditions, same messages
  if (d->n_splits < 0 || d->n_splits > 2) return fail(ABD_ERR_ARG, "only implemented 1-3 time chunks (0-2 splits)");
  for (int k = 0; k < d->n_splits; ++k)
    if (d->splits[k] < 0) return fail(ABD_ERR_ARG, "split indexes must be positive");
  if (d->n_splits == 2 && d->splits[0] > d->splits[1]) return fail(ABD_ERR_ARG, "splits must be in ascending order");
  if (d->n_splits > 0 && d->splits[d->n_splits - 1] > G) return fail(ABD_ERR_ARG, "largest split must be less than n_gaps - 1, (%d)", d->splits[d->n_splits - 1]);
  if (d->n_splits == 2 && d->splits[0] == d->splits[1]) return fail(ABD_ERR_ARG, "splits not unique");
  for (int64_t k = 0; k < (int64_t)G * N; ++k) {
    if ((d->vacs[k] != 0 && d->vacs[k] != 1)) return fail(ABD_ERR_ARG, "vacs must be 0/1");
    if (d->pcrpos && d->pcrpos[k] != 0 && d->pcrpos[k] != 1) return fail(ABD_ERR_ARG, "pcrpos must be 0/1");
  }

  SortedObs so_s, so_n;
  int rc = sort_obs(d->s, G, N, "s", so_s);
  if (rc) return rc;
  rc = sort_obs(d->n, G, N, "n", so_n);
  if (rc) return rc;

  abd_ctx* c = new (std::nothrow) abd_ctx();
  if (!c) return fail(ABD_ERR_NOMEM, "out of host memory");
  c->G = G;
  c->N = N;
  c->nt = (G + 63) / 64;
  c->prior_const = prior_constant(G);
  c->n_lg = (N + 63) / 64;
  c->n_chunks = d->n_splits + 1;
  c->storage = d->storage;
  c->dense = so_s.one_per_cell && so_n.one_per_cell;
  // the dense kernel addresses the <= 34 gap rows of a chunk with a 32-bit scalar offset (abd_dense.hpp); beyond
  // ~8 M individuals per GPU the cohort takes the observation-list kernels instead
  if ((int64_t)N * (d->storage == ABD_STORE_F32 ? 8 : 16) * 34 >= ((int64_t)1 << 32)) c->dense = false;
  if (const char* e = std::getenv("ABD_FORCE_SPARSE"))
    if (std::atoi(e)) c->dense = false;
  c->ignore_pcr = d->pcrpos == nullptr;
  c->n_slots = d->n_chain_slots;
  {
    const int edges[4] = {0, d->n_splits > 0 ? d->splits[0] : G, d->n_splits > 1 ? d->splits[1] : G, G};
    for (int ch = 0; ch < c->n_chunks; ++ch) {
      const int lo = edges[ch], hi = (ch == c->n_chunks - 1) ? G : edges[ch + 1];
      for (int g = lo; g < hi; ++g) c->chunk_mask[ch][g >> 6] |= 1ull << (g & 63);
    }
  }

#define CREATE_TRY(expr)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      fail(ABD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                                    \
      free_ctx(c);                                                                                  \
      return ABD_ERR_HIP;                                                                           \
    }                                                                                               \
  } while (0)

  int dev = d->device;
  if (dev < 0) CREATE_TRY(hipGetDevice(&dev));
  c->device = dev;
  CREATE_TRY(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CREATE_TRY(hipGetDeviceProperties(&prop, dev));
  c->n_cu = prop.multiProcessorCount;
  snprintf(c->name, sizeof c->name, "%s %s %d CUs", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  // sparse kernel: persistent waves, one individual at a time
  int bpc = 2;
  if (const char* e = std::getenv("ABD_BLOCKS_PER_CU")) bpc = std::max(1, std::atoi(e));
  const int sparse_max = std::max(1, std::min((N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 16));
  c->blocks_x = std::max(1, std::min(sparse_max, c->n_cu * bpc));
  // dense kernel: 4 workgroups per CU = 4 waves per SIMD (<= 128 VGPRs, ~29 KB LDS each): one round, equal ranges
  int dbpc = 4;
  if (const char* e = std::getenv("ABD_DENSE_BLOCKS_PER_CU")) dbpc = std::max(1, std::atoi(e));
  {
    const int cap = c->n_cu * 8;
    c->ob_n = (int)std::min<int64_t>((d->n.n_obs + ABD_BLOCK - 1) / ABD_BLOCK, cap);
    c->ob_s = (int)std::min<int64_t>((d->s.n_obs + ABD_BLOCK - 1) / ABD_BLOCK, cap);
    c->ob_c = std::max(1, std::min((N + ABD_BLOCK - 1) / ABD_BLOCK, 64));
    // lane per observation unless the lists are so full that a wave per individual keeps its 64 lanes busy
    // for two rounds or more and amortises the constraint pass (measured crossover, tools/bench_sparse.py)
    c->obs_lanes = d->s.n_obs + d->n.n_obs < (int64_t)256 * N;
    if (const char* e = std::getenv("ABD_OBS_LANES")) c->obs_lanes = std::atoi(e) != 0;
  }
  c->blocks_max = std::max({sparse_max, c->n_cu * 16, c->ob_n + c->ob_s + c->ob_c});
  c->dense_blocks = std::min(c->n_cu * dbpc, c->blocks_max);
  if (const char* e = std::getenv("ABD_CPW")) c->cpw_forced = std::atoi(e);
  if (table_lds_bytes(G, 4, 16, true) > 160 * 1024) {
    free_ctx(c);
    return fail(ABD_ERR_ARG, "LDS tables for n_gaps=%d do not fit", G);
  }

  if (c->storage == ABD_STORE_F32) {
    rc = upload_antigen<float>(c, d->s, so_s, c->s);
    if (!rc) rc = upload_antigen<float>(c, d->n, so_n, c->n);
  } else {
    rc = upload_antigen<double>(c, d->s, so_s, c->s);
    if (!rc) rc = upload_antigen<double>(c, d->n, so_n, c->n);
  }
  if (rc) {
    free_ctx(c);
    return rc;
  }
  const size_t cells = (size_t)G * N;
  const size_t words = (size_t)c->nt * N;
  {
    const std::vector<uint64_t> vw = pack_ng(d->vacs, G, N, c->nt);
    const std::vector<uint64_t> pw = pack_ng(d->pcrpos, G, N, c->nt);
    CREATE_TRY(hipMalloc(&c->vw, words * sizeof(uint64_t)));
    CREATE_TRY(hipMalloc(&c->pw, words * sizeof(uint64_t)));
    CREATE_TRY(hipMemcpy(c->vw, vw.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->pw, pw.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  if (c->dense) {
    // 2^(j/1024) rounded once from the 64-bit-mantissa value
    std::vector<double> tab(ABD_EXP2_TAB);
    for (int j = 0; j < ABD_EXP2_TAB; ++j) tab[(size_t)j] = (double)exp2l((long double)j / (long double)ABD_EXP2_TAB);
    CREATE_TRY(hipMalloc(&c->exp2_tab, tab.size() * sizeof(double)));
    CREATE_TRY(hipMemcpy(c->exp2_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  CREATE_TRY(hipMalloc(&c->stage_gn, cells));
  c->slots.resize((size_t)c->n_slots);
  for (auto& s : c->slots) {
    CREATE_TRY(hipMalloc(&s.rw, words * sizeof(uint64_t)));
    CREATE_TRY(hipMalloc(&s.waner, (size_t)N));
  }
  c->pipe[0].st = c->stream;
  if (const char* e = std::getenv("ABD_TWO_PIPES")) c->n_pipes = std::atoi(e) != 0 ? 2 : 1;
  if (const char* e = std::getenv("ABD_PIPES")) c->n_pipes = std::max(1, std::min(6, std::atoi(e)));
  if (!c->dense) c->n_pipes = 1;  // only the dense kernel has a grid for sharing the chip; the others just overlap
  c->n_streams = kMaxPipes;
  c->n_sync_slots = std::max(4, c->n_slots);
  for (int pi = 1; pi < c->n_streams; ++pi) {
    CREATE_TRY(hipStreamCreateWithFlags(&c->pipe[pi].st, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&c->join_ev[pi], hipEventDisableTiming));
  }
  // a launch that shares the chip with the other pipes' launches gets 1/n_pipes of the workgroup slots: fewer,
  // longer ranges, i.e. less per-range set-up for the same work
  c->pipe_blocks = std::min(c->dense_blocks, c->n_cu * std::max(1, dbpc / c->n_pipes));  // measured best: 3 pipes x 1 workgroup per CU
  if (const char* e = std::getenv("ABD_PIPE_BLOCKS_PER_CU")) c->pipe_blocks = std::max(1, std::min(c->n_cu * std::atoi(e), c->blocks_max));
  if (const char* e = std::getenv("ABD_PIPE_BLOCKS")) c->pipe_blocks = std::max(1, std::min(std::atoi(e), c->blocks_max));
  c->dbpc = dbpc;
  c->group_blocks = std::min(c->dense_blocks, c->n_cu * std::max(1, dbpc / 2));
  for (int pi = 0; pi < kMaxPipes; ++pi)
    if (c->pipe[pi].st)
      for (int b = 0; b < 2; ++b)
        CREATE_TRY(hipMalloc(&c->pipe[pi].partials[b], (size_t)c->n_slots * c->blocks_max * ABD_NOUT * sizeof(double)));
  if (const char* e = std::getenv("ABD_FUSE_FINALIZE")) c->fuse_finalize = std::atoi(e) != 0;
  if (const char* e = std::getenv("ABD_XCD_REMAP")) c->xcd_remap = std::atoi(e) != 0;
  if (const char* e = std::getenv("ABD_FIN_ROWS")) c->fin_rows = std::max(0, std::atoi(e));
  const size_t out_bytes = (size_t)(kResultSlots + c->n_sync_slots) * c->n_slots * ABD_NOUT * sizeof(double);
  // COHERENT (fine-grained) on purpose: synchronous calls poll a completion tag in this memory while the stream
  // is still running.  With hipHostMallocMapped alone the allocation is non-coherent: the GPU caches it and the
  // two 64-byte halves of a result row could reach the host in either order (tag visible, data stale).
  CREATE_TRY(hipHostMalloc(&c->h_out, out_bytes, hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(c->h_out, 0, out_bytes);
  CREATE_TRY(hipHostGetDevicePointer((void**)&c->d_out, c->h_out, 0));
  CREATE_TRY(hipHostMalloc(&c->h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(c->h_done, 0, 64);
  CREATE_TRY(hipHostGetDevicePointer((void**)&c->d_done, c->h_done, 0));
  if (const char* e = std::getenv("ABD_WAIT_POLL")) c->wait_poll = std::atoi(e) != 0;
  if (const char* e = std::getenv("ABD_DIRECT_OUT")) c->direct_out = std::atoi(e) != 0;
  CREATE_TRY(hipMalloc(&c->d_ring, out_bytes));
  CREATE_TRY(hipMalloc(&c->d_counts, ((size_t)c->n_slots * 2 + 8) * sizeof(unsigned long long)));  // + 8 development counters
  CREATE_TRY(hipMalloc(&c->d_work, (size_t)2 * c->n_slots * sizeof(unsigned int)));
  CREATE_TRY(hipMalloc(&c->d_counts_chain, (size_t)c->n_slots * 2 * sizeof(unsigned long long)));
  CREATE_TRY(hipMalloc(&c->d_fin_count, (size_t)kMaxPipes * ABD_MAX_BATCH * sizeof(unsigned int)));
  CREATE_TRY(hipMemset(c->d_fin_count, 0, (size_t)kMaxPipes * ABD_MAX_BATCH * sizeof(unsigned int)));
  if (const char* e = std::getenv("ABD_OBS_FUSED_SUM")) c->obs_fused = std::atoi(e) != 0;
  if (const char* e = std::getenv("ABD_DENSE_OWN_SUM")) c->dense_own_sum = std::atoi(e) != 0;
  CREATE_TRY(hipHostMalloc(&c->h_counts_chain, (size_t)c->n_slots * 2 * sizeof(unsigned long long), hipHostMallocDefault));
  if (const char* e = std::getenv("ABD_GIBBS_V1")) c->gibbs_v1 = std::atoi(e) != 0;
  if (const char* e = std::getenv("ABD_G2_REFILL_MIN")) c->g2_refill_min = std::max(1, std::min(64, std::atoi(e)));
  if (const char* e = std::getenv("ABD_G2_TAIL_LANES")) c->g2_tail_lanes = std::max(0, std::min(64, std::atoi(e)));
  if (const char* e = std::getenv("ABD_G2_TAIL_AGE")) c->g2_tail_age = std::max(0, std::atoi(e));
  c->results.resize((size_t)kResultSlots + c->n_sync_slots);
  CREATE_TRY(hipStreamSynchronize(c->stream));
  if (c->dense && c->n_pipes > 1) {
    // the pipes must sit on different hardware queues (two launches in one queue run one after the other: 4 pipes on
    // streams 0..3, two of which share a queue here, gave 143 k evals/s at config 3 against 192 k on streams 0, 1, 2, 5)
    if (int prc = probe_stream_queues(c)) {
      free_ctx(c);
      return prc;
    }
    c->n_pipes = std::min(c->n_pipes, c->n_queues);
  }
#undef CREATE_TRY
  *out = c;
  return ABD_OK;
}

int abd_destroy(abd_ctx* c) {
  free_ctx(c);
  return ABD_OK;
}

int abd_device_name(abd_ctx* c, char* buf, int32_t buflen) {
  if (!c || !buf || buflen < 1) return fail(ABD_ERR_ARG, "bad argument");
  snprintf(buf, (size_t)buflen, "%s", c->name);
  return ABD_OK;
}

int abd_is_dense(abd_ctx* c) { return c && c->dense ? 1 : 0; }
int abd_n_pipes(abd_ctx* c) { return c ? c->n_pipes : -1; }

int abd_set_discrete(abd_ctx* c, int32_t chain, const int8_t* i_raw, const int8_t* waner) {
  if (!c || !i_raw || !waner) return fail(ABD_ERR_ARG, "NULL argument");
  if (chain < 0 || chain >= c->n_slots) return fail(ABD_ERR_ARG, "chain %d outside [0, %d)", chain, c->n_slots);
  const size_t cells = (size_t)c->G * c->N;
  for (size_t k = 0; k < cells; ++k)
    if (i_raw[k] != 0 && i_raw[k] != 1) return fail(ABD_ERR_ARG, "i_raw must be 0/1");
  for (int j = 0; j < c->N; ++j)
    if (waner[j] != 0 && waner[j] != 1) return fail(ABD_ERR_ARG, "ab_s_waner must be 0/1");
  HIP_TRY(hipSetDevice(c->device));
  ChainSlot& s = c->slots[(size_t)chain];
  // synchronous copies: the caller's buffers may be reused immediately
  if (int jrc = join_pipes(c)) return jrc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(c->stage_gn, i_raw, cells, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s.waner, waner, (size_t)c->N, hipMemcpyHostToDevice));
  dim3 grid((c->N + 255) / 256, c->nt);
  hipLaunchKernelGGL(abd_pack_bits_kernel, grid, dim3(256), 0, c->stream, c->stage_gn, s.rw, c->G, c->N, c->nt);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  s.set = true;
  return ABD_OK;
}

int abd_flip_discrete(abd_ctx* c, int32_t chain, int64_t flat) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (chain < 0 || chain >= c->n_slots) return fail(ABD_ERR_ARG, "chain %d outside [0, %d)", chain, c->n_slots);
  const int64_t total = (int64_t)c->G * c->N + c->N;
  if (flat < 0 || flat >= total) return fail(ABD_ERR_ARG, "flat index %lld outside [0, %lld)", (long long)flat, (long long)total);
  if (!c->slots[(size_t)chain].set) return fail(ABD_ERR_STATE, "chain slot %d has no discrete state", chain);
  HIP_TRY(hipSetDevice(c->device));
  if (int jrc = join_pipes(c)) return jrc;
  ChainSlot& s = c->slots[(size_t)chain];
  hipLaunchKernelGGL(abd_flip_kernel, dim3(1), dim3(1), 0, c->stream, s.rw, s.waner, c->G, c->N, flat);
  HIP_TRY(hipGetLastError());
  return ABD_OK;
}

int abd_n_result_slots(abd_ctx*) { return kResultSlots; }

int abd_logp_dlogp_batch_enqueue(abd_ctx* c, int32_t slot, int32_t n, const int32_t* chains, const double* theta) {
  if (!c || !chains || !theta) return fail(ABD_ERR_ARG, "NULL argument");
  if (slot < 0 || slot >= kResultSlots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slot, kResultSlots);
  return enqueue_slot(c, slot, n, chains, theta, true, true);
}

int abd_wait(abd_ctx* c) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  c->flush_tag = 0.0;  // only a flush queued by THIS call may stand in for the synchronise
  int rc = flush_ring(c);
  if (rc) return rc;
  if (c->direct_out && !c->pending_slots.empty() && c->wait_poll) {
    // every stream-ordered slot's rows carry a tag: the newest slots land last, so poll backwards and stop early
    for (size_t q = c->pending_slots.size(); q-- > 0;)
      if (int wrc = wait_slot(c, c->pending_slots[q])) return wrc;
    c->pending_slots.clear();
    return ABD_OK;
  }
  c->pending_slots.clear();
  if (c->flush_tag != 0.0) {  // the flush raises a tag in mapped memory when it is done
    const double tag = c->flush_tag;
    c->flush_tag = 0.0;
    volatile const double* done = c->h_done;
    const auto t_poll = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
      if (*done == tag) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return ABD_OK;
      }
      __builtin_ia32_pause();
      // a long queue is not a fault: after a second of polling hand over to the (always correct) stream synchronise
      if ((spin & 4095) == 4095 && std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(1)) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ABD_OK;
}

int abd_fetch(abd_ctx* c, int32_t slot, double* logp, double* grad) {
  if (!c || !logp) return fail(ABD_ERR_ARG, "NULL argument");
  if (slot < 0 || slot >= kResultSlots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slot, kResultSlots);
  return fetch_slot(c, slot, logp, grad);
}

int abd_fetch_many(abd_ctx* c, int32_t n_slots, const int32_t* slots, double* logp, double* grad) {
  if (!c || !slots || !logp) return fail(ABD_ERR_ARG, "NULL argument");
  size_t off = 0;
  for (int s = 0; s < n_slots; ++s) {
    if (slots[s] < 0 || slots[s] >= kResultSlots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slots[s], kResultSlots);
    const int n = c->results[slots[s]].n;
    int rc = fetch_slot(c, slots[s], logp + off, grad ? grad + off * ABD_N_THETA : nullptr);
    if (rc) return rc;
    off += (size_t)n;
  }
  return ABD_OK;
}

int abd_logp_dlogp_many(abd_ctx* c, int32_t n_steps, int32_t n, const int32_t* chains, const double* theta, double* logp,
                        double* grad) {
  if (!c || !chains || !theta || !logp) return fail(ABD_ERR_ARG, "NULL argument");
  if (n_steps < 0) return fail(ABD_ERR_ARG, "n_steps=%d is negative", n_steps);
  const size_t per_step = (size_t)n * ABD_N_THETA;
  for (int s0 = 0; s0 < n_steps; s0 += kResultSlots) {  // windows of the result ring
    const int s1 = std::min(n_steps, s0 + kResultSlots);
    for (int k = s0; k < s1; ++k)
      if (int rc = enqueue_slot(c, k - s0, n, chains, theta + (size_t)k * per_step, grad != nullptr, true)) return rc;
    if (c->direct_out && c->wait_poll) {
      // the results land in mapped host memory slot by slot: queue the pending sums and the joins, then take every step's
      // result as soon as its tag shows -- the host-side assembly of the early steps overlaps the late steps' kernels
      HIP_TRY(hipSetDevice(c->device));
      c->flush_tag = 0.0;
      if (int rc = flush_ring(c)) return rc;
      for (int k = s0; k < s1; ++k) {
        if (int rc = wait_slot(c, k - s0)) return rc;
        if (int rc = fetch_slot(c, k - s0, logp + (size_t)k * n, grad ? grad + (size_t)k * per_step : nullptr)) return rc;
      }
      c->pending_slots.clear();
      continue;
    }
    if (int rc = abd_wait(c)) return rc;
    for (int k = s0; k < s1; ++k)
      if (int rc = fetch_slot(c, k - s0, logp + (size_t)k * n, grad ? grad + (size_t)k * per_step : nullptr)) return rc;
  }
  return ABD_OK;
}

int abd_logp_dlogp_batch(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta, double* logp, double* grad) {
  if (!c || !chains || !theta || !logp || !grad) return fail(ABD_ERR_ARG, "NULL argument");
  if (int frc = flush_ring(c)) return frc;
  int rc = enqueue_slot(c, kSyncSlot, n, chains, theta, true);
  if (rc) return rc;
  if (int prc = flush_pending(c)) return prc;  // a synchronous call sums its own partials right away
  if (int wrc = wait_rows(c, kSyncSlot, c->results[kSyncSlot].n, c->seq)) return wrc;
  return fetch_slot(c, kSyncSlot, logp, grad);
}

int abd_logp_dlogp(abd_ctx* c, int32_t chain, const double* theta, double* logp, double* grad) {
  return abd_logp_dlogp_batch(c, 1, &chain, theta, logp, grad);
}

int abd_loglik_dlogp(abd_ctx* c, int32_t chain, const double* theta, double* loglik, double* grad) {
  if (!c || !theta || !loglik || !grad) return fail(ABD_ERR_ARG, "NULL argument");
  int rc = enqueue_slot(c, kSyncSlot, 1, &chain, theta, true);
  if (rc) return rc;
  if (int prc = flush_pending(c)) return prc;  // a synchronous call sums its own partials right away
  if (int wrc = wait_rows(c, kSyncSlot, c->results[kSyncSlot].n, c->seq)) return wrc;
  return fetch_slot(c, kSyncSlot, loglik, grad, false);
}

int abd_logp(abd_ctx* c, int32_t chain, const double* theta, double* logp) {
  if (!c || !theta || !logp) return fail(ABD_ERR_ARG, "NULL argument");
  int rc = enqueue_slot(c, kSyncSlot, 1, &chain, theta, false);
  if (rc) return rc;
  if (int prc = flush_pending(c)) return prc;  // a synchronous call sums its own partials right away
  if (int wrc = wait_rows(c, kSyncSlot, c->results[kSyncSlot].n, c->seq)) return wrc;
  return fetch_slot(c, kSyncSlot, logp, nullptr);
}

int abd_deterministics(abd_ctx* c, int32_t chain, const double* theta, int8_t* i, double* mu_n, double* mu_s) {
  if (!c || !theta) return fail(ABD_ERR_ARG, "NULL argument");
  int rc = check_chains(c, 1, &chain);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if (int jrc = join_pipes(c)) return jrc;
  const size_t cells = (size_t)c->G * c->N;
  if (!c->d_det) HIP_TRY(hipMalloc(&c->d_det, cells * (2 * sizeof(double) + 1)));  // staging, kept for the next draw
  double* d_n = c->d_det;
  double* d_s = c->d_det + cells;
  int8_t* d_i = reinterpret_cast<int8_t*>(c->d_det + 2 * cells);
  EvalArgs a;
  base_args(c, a);
  a.n_chains = 1;
  a.ch[0] = chain_par(c, chain, theta);
  const size_t lds = (size_t)3 * (c->G + 1) * sizeof(double2_t);
  const int blocks = std::max(1, std::min((c->N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 8));
  hipLaunchKernelGGL(abd_deterministics_kernel, dim3(blocks), dim3(ABD_BLOCK), lds, c->stream, a, i ? d_i : (int8_t*)nullptr,
                     mu_n ? d_n : (double*)nullptr, mu_s ? d_s : (double*)nullptr, (double*)nullptr);
  HIP_TRY(hipGetLastError());
  if (mu_n) HIP_TRY(hipMemcpyAsync(mu_n, d_n, cells * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (mu_s) HIP_TRY(hipMemcpyAsync(mu_s, d_s, cells * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (i) HIP_TRY(hipMemcpyAsync(i, d_i, cells, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ABD_OK;
}

// Queue one sweep launch for m <= ABD_MAX_BATCH chains on stream st (nothing is waited for): counts of chain k of the
// launch go to counts_dev[2 k .. 2 k + 1] (zeroed first), the dense kernel's work queues are work_dev[0 .. m).
static int enqueue_gibbs(abd_ctx* c, int m, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
                         uint32_t stream_offset, hipStream_t st, unsigned long long* counts_dev, unsigned int* work_dev,
                         unsigned long long* stats_dev) {
  GibbsArgs ga;
  base_args(c, ga.e);
  ga.e.n_chains = m;
  ga.seed_lo = (uint32_t)seed;
  ga.seed_hi = (uint32_t)(seed >> 32);
  ga.sweep = sweep;
  ga.ind_offset = c->ind_offset;
  ga.counts = counts_dev;
  for (int k = 0; k < m; ++k) {
    const double* t = theta + (size_t)k * ABD_N_THETA;
    ga.e.ch[k] = chain_par(c, chains[k], t);
    const Transformed tr = transform(t);
    ga.stream[k] = (uint32_t)chains[k] + stream_offset;
    ga.theta0[k] = t[0];
    ga.theta7[k] = t[7];
    ga.is2_n[k] = 1.0 / (tr.sig_n * tr.sig_n);
    ga.is2_s[k] = 1.0 / (tr.sig_s * tr.sig_s);
  }
  HIP_TRY(hipMemsetAsync(counts_dev, 0, (size_t)m * 2 * sizeof(unsigned long long), st));
  ga.work = work_dev;
  ga.refill_min = c->g2_refill_min;
  ga.tail_lanes = c->g2_tail_lanes;
  ga.tail_age = c->g2_tail_age;
  ga.stats = stats_dev;
  if (stats_dev) HIP_TRY(hipMemsetAsync(stats_dev, 0, 8 * sizeof(unsigned long long), st));
  const int rbytes = c->storage == ABD_STORE_F32 ? 4 : 8;
  const int nw2 = abd_g2_waves(c->G, rbytes);  // waves of a workgroup = of a CU: as many as its LDS holds, 12 at most
  if (c->dense && !c->gibbs_v1 && nw2 >= 4) {
    // lanes = proposals (abd_gibbs2.hpp): one workgroup per CU, the individuals of a chain handed out from one queue
    // per chain
    const size_t lds2 = abd_g2_lds(c->G, rbytes, nw2);
    HIP_TRY(hipMemsetAsync(work_dev, 0, (size_t)m * sizeof(unsigned int), st));
    const int bx = std::max(1, std::min(c->n_cu / m, (c->N + nw2 - 1) / nw2));
    dim3 grid2(bx, m);
    if (c->storage == ABD_STORE_F32) {
      if (lds2 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(abd_gibbs_dense_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      hipLaunchKernelGGL((abd_gibbs_dense_kernel<float>), grid2, dim3(64 * nw2), lds2, st, ga);
    } else {
      if (lds2 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(abd_gibbs_dense_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      hipLaunchKernelGGL((abd_gibbs_dense_kernel<double>), grid2, dim3(64 * nw2), lds2, st, ga);
    }
  } else {
    const int blocks = std::max(1, std::min((c->N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 8));
    const size_t lds = (size_t)3 * (c->G + 1) * sizeof(double2_t) + (size_t)ABD_WAVES_PER_BLOCK * ABD_GIBBS_WAVE_LDS;
    dim3 grid(blocks, m);
    if (c->dense) {
      if (c->storage == ABD_STORE_F32)
        hipLaunchKernelGGL((abd_gibbs_kernel<float, true>), grid, dim3(ABD_BLOCK), lds, st, ga);
      else
        hipLaunchKernelGGL((abd_gibbs_kernel<double, true>), grid, dim3(ABD_BLOCK), lds, st, ga);
    } else {
      if (c->storage == ABD_STORE_F32)
        hipLaunchKernelGGL((abd_gibbs_kernel<float, false>), grid, dim3(ABD_BLOCK), lds, st, ga);
      else
        hipLaunchKernelGGL((abd_gibbs_kernel<double, false>), grid, dim3(ABD_BLOCK), lds, st, ga);
    }
  }
  HIP_TRY(hipGetLastError());
  return ABD_OK;
}

static int gibbs_sweep_impl(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
                            uint32_t stream_offset, int64_t* accepted, int64_t* proposed) {
  if (!c || !chains || !theta) return fail(ABD_ERR_ARG, "NULL argument");
  int rc = check_chains(c, n, chains);
  if (rc) return rc;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (chains[a] == chains[b]) return fail(ABD_ERR_ARG, "chain %d listed twice: a sweep updates its state in place", chains[a]);
  HIP_TRY(hipSetDevice(c->device));
  if (int frc = flush_ring(c)) return frc;
  std::vector<unsigned long long> counts((size_t)n * 2, 0);
  static const bool want_stats = std::getenv("ABD_GIBBS_STATS") && std::atoi(std::getenv("ABD_GIBBS_STATS")) != 0;
  for (int k0 = 0; k0 < n; k0 += ABD_MAX_BATCH) {
    const int m = std::min(ABD_MAX_BATCH, n - k0);
    unsigned long long* stats_dev = want_stats ? c->d_counts + (size_t)c->n_slots * 2 : nullptr;
    rc = enqueue_gibbs(c, m, chains + k0, theta + (size_t)k0 * ABD_N_THETA, seed, sweep, stream_offset, c->stream, c->d_counts,
                       c->d_work, stats_dev);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(counts.data() + (size_t)k0 * 2, c->d_counts, (size_t)m * 2 * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats_dev) {
      unsigned long long st[8];
      HIP_TRY(hipMemcpy(st, stats_dev, sizeof st, hipMemcpyDeviceToHost));
      const double ni = (double)std::max<unsigned long long>(1, st[0]);
      std::fprintf(stderr, "[abd gibbs stats] individuals x chains %llu; per individual: iterations %.1f, refills %.1f, walk steps %.1f "
                   "(lanes busy %.1f of 64), tail finishes %.1f, commit scans %.1f, acceptances %.2f\n",
                   st[0], st[1] / ni, st[2] / ni, st[3] / ni, st[3] ? (double)st[4] / (double)st[3] : 0.0, st[5] / ni, st[6] / ni, st[7] / ni);
    }
  }
  for (int k = 0; k < n; ++k) {
    if (accepted) accepted[k] = (int64_t)counts[(size_t)k * 2];
    if (proposed) proposed[k] = (int64_t)counts[(size_t)k * 2 + 1];
  }
  return ABD_OK;
}

int abd_gibbs_sweep(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
                    int64_t* accepted, int64_t* proposed) {
  return gibbs_sweep_impl(c, n, chains, theta, seed, sweep, 0u, accepted, proposed);
}

int abd_get_discrete(abd_ctx* c, int32_t chain, int8_t* i_raw, int8_t* waner) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  int rc = check_chains(c, 1, &chain);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if (int jrc = join_pipes(c)) return jrc;
  ChainSlot& s = c->slots[(size_t)chain];
  if (i_raw) {
    dim3 grid((c->N + 255) / 256, c->G);
    hipLaunchKernelGGL(abd_unpack_bits_kernel, grid, dim3(256), 0, c->stream, s.rw, c->stage_gn, c->G, c->N);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(i_raw, c->stage_gn, (size_t)c->G * c->N, hipMemcpyDeviceToHost));
  } else {
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (waner) HIP_TRY(hipMemcpy(waner, s.waner, (size_t)c->N, hipMemcpyDeviceToHost));
  return ABD_OK;
}

int abd_set_launch_config(abd_ctx* c, int32_t blocks, int32_t chains_per_wave) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (!(chains_per_wave == 0 || chains_per_wave == 1 || chains_per_wave == 2 || chains_per_wave == 4))
    return fail(ABD_ERR_ARG, "chains_per_wave must be 0 (auto), 1, 2 or 4");
  c->cpw_forced = chains_per_wave;
  if (blocks > 0) {
    if (c->dense)
      c->dense_blocks = std::max(1, std::min(blocks, c->blocks_max));
    else
      c->blocks_x = std::max(1, std::min(blocks, c->blocks_max));
  }
  return ABD_OK;
}

int abd_theta_prior(abd_ctx* c, const double* theta, double* logp, double* grad) {
  if (!c || !theta || !logp) return fail(ABD_ERR_ARG, "NULL argument");
  *logp = priors(prepare(theta), theta, c->G, 0.0, 0.0, 0.0, 0.0, grad, c->prior_const);
  return ABD_OK;
}

int abd_set_individual_offset(abd_ctx* c, int64_t first_individual) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (first_individual < 0 || first_individual > 0xFFFFFFFFll) return fail(ABD_ERR_ARG, "first_individual=%lld out of range", (long long)first_individual);
  c->ind_offset = (uint32_t)first_individual;
  return ABD_OK;
}

int abd_kernel_timing(abd_ctx* c, int32_t mode) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (mode < 0 || mode > 2) return fail(ABD_ERR_ARG, "timing mode %d outside {0, 1, 2}", mode);
  if (mode != c->timing) {
    HIP_TRY(hipSetDevice(c->device));
    if (int frc = flush_ring(c)) return frc;  // closes an open window, joins the pipes
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  c->timing = mode;
  return ABD_OK;
}

int abd_kernel_time(abd_ctx* c, double* total_ms, int64_t* launches, int32_t reset) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  if (int frc = flush_ring(c)) return frc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < c->ev_used; ++k) {
    float ms = 0.f;
    if (c->timing == 2) {  // window: first launch's start .. the latest pipe's end
      for (int pi = 0; pi < c->n_streams; ++pi) {
        if (!c->pipe[pi].st || (k + 1) * kMaxPipes > c->win_end.size() || k >= c->win_mask.size() || !(c->win_mask[k] >> pi & 1u)) continue;
        float m = 0.f;
        if (hipEventElapsedTime(&m, c->ev_pool[k].first, c->win_end[k * kMaxPipes + pi]) == hipSuccess) ms = std::max(ms, m);
      }
    } else {
      HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[k].first, c->ev_pool[k].second));
    }
    c->ev_total_ms += ms;
    if (c->timing != 2) c->ev_count++;
  }
  if (c->timing == 2) {
    c->ev_count += c->win_launches;
    c->win_launches = 0;
  }
  c->ev_used = 0;
  if (total_ms) *total_ms = c->ev_total_ms;
  if (launches) *launches = c->ev_count;
  if (reset) {
    c->ev_total_ms = 0.0;
    c->ev_count = 0;
  }
  return ABD_OK;
}

int64_t abd_wait_fallbacks(abd_ctx* c) { return c ? c->wait_fallbacks : -1; }
int abd_stream_queues(abd_ctx* c, int32_t* queue_of_stream, int32_t n) {
  if (!c || !queue_of_stream) return fail(ABD_ERR_ARG, "NULL argument");
  if (int rc = probe_stream_queues(c)) return rc;
  for (int i = 0; i < n; ++i) queue_of_stream[i] = i < c->n_streams ? c->queue_of_pipe[i] : -1;
  return ABD_OK;
}
int abd_resident_stats(abd_ctx* c, int64_t* launches, int64_t* commands, int64_t* restarts) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (launches) *launches = c->resident_launches;
  if (commands) *commands = c->resident_commands;
  if (restarts) *restarts = c->resident_restarts;
  return ABD_OK;
}

int64_t abd_algorithmic_bytes(abd_ctx* c, int32_t n_chains) {
  if (!c) return 0;
  const int64_t R = c->storage == ABD_STORE_F32 ? 4 : 8;
  const int64_t cells = (int64_t)c->G * c->N;
  // indicator panels are bit-packed in 64-gap words: vacs + pcrpos + one i_raw per chain, plus the waner bytes
  const int64_t bits = (int64_t)c->nt * c->N * 8 * (2 + n_chains) + (int64_t)n_chains * c->N;
  if (c->dense) return cells * 4 * R + bits;
  return (c->s.K + c->n.K) * (2 * R + 1) + 2 * (int64_t)(c->N + 1) * 4 + bits;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Native compound sampler: lock-step NUTS over the chains + the device Gibbs sweep (abd_hip.h)
// ---------------------------------------------------------------------------------------------------
static int gibbs_sweep_impl(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
                            uint32_t stream_offset, int64_t* accepted, int64_t* proposed);

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
  // dense cohorts, one chain per unit: the evaluation kernel stays resident for a whole trajectory (abd_resident.hpp)
  struct Resident {
    unsigned long long* mail_h = nullptr;  // mapped host memory: two mailboxes of ABD_RES_WORDS words, used alternately per launch
    unsigned long long* mail_d = nullptr;  // ... as the device sees them
    unsigned int* status_h = nullptr;      // behind the mailboxes: [0] how the kernel ended, [1] commands served
    unsigned int* status_d = nullptr;
    unsigned long long* relay = nullptr;   // device: ABD_RES_RELAY_WORDS words, then the `done` counter
    double* partials = nullptr;            // device: [blocks][ABD_NOUT]
    unsigned long long rounds = 0;         // commands served by this unit's kernels so far (done counter = rounds * blocks)
    int cur = 0;
    bool live = false;
    int fails = 0;  // relaunches in a row without an answer
  };
  std::vector<Resident> res;
  // one completion-tag sequence per unit, disjoint from the context's and from each other's (unit u: (u + 1) 2^40 + k)
  std::vector<double> unit_seq;
  int threads = 1;  // host threads that drive the units (sampler_run_units)
  bool resident = false;
  int res_blocks = 0;
  size_t res_lds = 0;
  unsigned long long res_cmd_timeout = 0, res_guard_timeout = 0;
};

namespace {

// add chain k's Deterministics at its current point to its running sums (stream st)
int accumulate_chain(abd_sampler* s, int k, hipStream_t st) {
  abd_ctx* c = s->c;
  const size_t cells = (size_t)c->G * c->N;
  const size_t lds = (size_t)3 * (c->G + 1) * sizeof(double2_t);
  const int blocks = std::max(1, std::min((c->N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 8));
  EvalArgs a;
  base_args(c, a);
  a.n_chains = 1;
  a.ch[0] = chain_par(c, s->chains[(size_t)k], s->ch[(size_t)k].nuts.q);
  hipLaunchKernelGGL(abd_deterministics_kernel, dim3(blocks), dim3(ABD_BLOCK), lds, st, a, (int8_t*)nullptr,
                     (double*)nullptr, (double*)nullptr, s->d_sums + (size_t)k * 3 * cells);
  HIP_TRY(hipGetLastError());
  return ABD_OK;
}

}  // namespace

namespace {

// ---- resident evaluation kernel of a sampler unit (abd_resident.hpp) ----

size_t resident_lds_bytes(int G) {
  return abd_res_tables_bytes(G) + (size_t)ABD_WAVES_PER_BLOCK * ABD_NOUT * sizeof(double) +
         (size_t)ABD_EXP2_TAB * sizeof(double) + (size_t)ABD_RES_WORDS * sizeof(unsigned long long) +
         (size_t)ABD_RES_PIECES * ABD_MAXT * ABD_BLOCK * sizeof(uint64_t) + 16;
}

template <typename R>
hipError_t resident_occupancy(int* per_cu, size_t lds) {
  const void* k = reinterpret_cast<const void*>(abd_dense_resident_kernel<R>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, abd_dense_resident_kernel<R>, ABD_BLOCK, lds);
}

void resident_free(abd_sampler* s) {
  for (auto& r : s->res) {
    if (r.mail_h) (void)hipHostFree(r.mail_h);
    if (r.relay) (void)hipFree(r.relay);
    if (r.partials) (void)hipFree(r.partials);
    r = abd_sampler::Resident();
  }
  s->res.clear();
  s->resident = false;
}

// Decide whether the units of this sampler keep their evaluation kernel resident, and allocate what that needs.
// Needs: a dense cohort, one chain per unit, every range of the unit's launch shape made of at most two pieces, and
// room on the chip for ALL units' workgroups at once (a resident workgroup that cannot start would be waited for).
int resident_setup(abd_sampler* s) {
  abd_ctx* c = s->c;
  s->resident = false;
  // Opt-in (ABD_RESIDENT=1): measured at config 3, trajectories without the sweep run 10-25 % faster,
  // but with the sweep the iteration gets SLOWER (2.2 -> 2.8 ms with 4 chains): a sweep workgroup (768 threads, 115 KB
  // of LDS) does not fit on a CU next to the other chains' resident workgroups, so every sweep waits for all the other
  // trajectories to end.
  const char* e = std::getenv("ABD_RESIDENT");
  if (!e || std::atoi(e) == 0) return ABD_OK;
  if (!c->dense || s->unit != 1) return ABD_OK;
  const int n_units = s->n;
  const int blocks = dense_blocks(c, 1, 2, 1);
  const size_t lds = resident_lds_bytes(c->G);
  int per_cu = 0;
  hipError_t oe = c->storage == ABD_STORE_F32 ? resident_occupancy<float>(&per_cu, lds) : resident_occupancy<double>(&per_cu, lds);
  if (oe != hipSuccess) return fail(ABD_ERR_HIP, "resident kernel occupancy: %s", hipGetErrorString(oe));
  if ((int64_t)n_units * blocks > (int64_t)per_cu * c->n_cu) return ABD_OK;
  if (n_units > 4 && !(e && std::atoi(e) >= 2)) return ABD_OK;  // more concurrent kernels than hardware queues (ABD_RESIDENT=2 tries anyway)
  const int32_t* tab = nullptr;
  if (int rrc = range_table(c, blocks, ABD_WAVES_PER_BLOCK, &tab)) return rrc;
  {  // at most two pieces per range: rows <= (G - first gap) + G
    std::vector<int32_t> h((size_t)blocks * ABD_WAVES_PER_BLOCK * 4);
    HIP_TRY(hipMemcpy(h.data(), tab, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < h.size() / 4; ++r)
      if (h[r * 4 + 2] > (c->G - h[r * 4 + 1]) + c->G) return ABD_OK;
  }
  s->res.resize((size_t)n_units);
  for (auto& r : s->res) {
    const size_t mail_bytes = 2 * ABD_RES_WORDS * sizeof(unsigned long long) + 128;
    HIP_TRY(hipHostMalloc((void**)&r.mail_h, mail_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(r.mail_h, 0, mail_bytes);
    HIP_TRY(hipHostGetDevicePointer((void**)&r.mail_d, r.mail_h, 0));
    r.status_h = reinterpret_cast<unsigned int*>(r.mail_h + 2 * ABD_RES_WORDS);
    r.status_d = reinterpret_cast<unsigned int*>(r.mail_d + 2 * ABD_RES_WORDS);
    const size_t relay_bytes = (size_t)(ABD_RES_RELAY_WORDS + 16) * sizeof(unsigned long long);
    HIP_TRY(hipMalloc(&r.relay, relay_bytes));
    HIP_TRY(hipMemset(r.relay, 0, relay_bytes));
    HIP_TRY(hipMalloc(&r.partials, (size_t)blocks * ABD_NOUT * sizeof(double)));
  }
  s->res_blocks = blocks;
  s->res_lds = lds;
  double ms = 20.0;  // without a command for this long the kernel leaves by itself (the host relaunches it)
  if (const char* t = std::getenv("ABD_RESIDENT_TIMEOUT_MS")) ms = std::max(0.001, std::atof(t));
  s->res_cmd_timeout = (unsigned long long)(ms * 1e5);  // s_memrealtime: 100 MHz
  s->res_guard_timeout = s->res_cmd_timeout * 10ull + 10000000ull;
  s->resident = true;
  s->threads = 1;  // the resident path shares the context's tag sequence
  return ABD_OK;
}

// the command: chain constants of theta, sequence number last in each 64-byte line (x86 stores become visible in order)
void resident_write(abd_sampler::Resident& r, const ChainPar& cp, double seq) {
  volatile unsigned long long* m = r.mail_h + (size_t)r.cur * ABD_RES_WORDS;
  const double lineA[7] = {cp.perm_n, cp.temp_n, cp.rho_n, cp.init_n, cp.perm_s, cp.rho_s, cp.init_s};
  const double lineB[4] = {cp.b_n, cp.d_n, cp.b_s, cp.d_s};
  unsigned long long w, sq;
  std::memcpy(&sq, &seq, sizeof sq);
  for (int k = 0; k < 7; ++k) {
    std::memcpy(&w, &lineA[k], sizeof w);
    m[k] = w;
  }
  __atomic_store_n(const_cast<unsigned long long*>(m) + 7, sq, __ATOMIC_RELEASE);
  for (int k = 0; k < 4; ++k) {
    std::memcpy(&w, &lineB[k], sizeof w);
    m[8 + k] = w;
  }
  __atomic_store_n(const_cast<unsigned long long*>(m) + 15, sq, __ATOMIC_RELEASE);
}

// Evaluate chain `chain` at theta through unit u's resident kernel (launched here if it is not running); the result
// lands in row 0 of slot kSyncSlot + u under the tag c->seq, like a launched evaluation's.
int resident_eval(abd_sampler* s, int u, int32_t chain, const double* theta) {
  abd_ctx* c = s->c;
  abd_sampler::Resident& r = s->res[(size_t)u];
  const int slot = kSyncSlot + u;
  ResultSlot& rs = c->results[slot];
  rs.n = 1;
  rs.grad = true;
  rs.chains.assign(1, chain);
  rs.theta.assign(theta, theta + ABD_N_THETA);
  rs.host.assign(1, prepare(theta));
  const ChainPar cp = chain_par(c, chain, rs.host[0].tr);
  if (r.live) {
    c->seq += 1.0;
    r.rounds += 1;
    c->resident_commands++;
    resident_write(r, cp, c->seq);
    return ABD_OK;
  }
  const int pi = unit_pipe(c, u);
  abd_ctx::Pipe& pp = c->pipe[pi];
  if (int frc = flush_pipe(c, pi)) return frc;
  if (pi > 0) pp.busy = true;
  ResidentArgs ra;
  base_args(c, ra.a);
  ra.a.n_chains = 1;
  ra.a.ch[0] = cp;
  if (int rrc = range_table(c, s->res_blocks, ABD_WAVES_PER_BLOCK, &ra.a.range_tab)) return rrc;
  ra.a.partials = r.partials;
  ra.a.fin_rows = c->fin_rows;
  ra.a.xcd_remap = c->xcd_remap ? 1 : 0;
  ra.mail = r.mail_d + (size_t)r.cur * ABD_RES_WORDS;
  ra.relay = r.relay;
  ra.done = r.relay + ABD_RES_RELAY_WORDS;
  ra.done0 = r.rounds * (unsigned long long)s->res_blocks;
  ra.out = c->d_out + (size_t)slot * c->n_slots * ABD_NOUT;
  ra.status = r.status_d;
  c->seq += 1.0;  // whatever an earlier kernel of this unit left in the relay words is older than this
  ra.seq0 = c->seq;
  ra.cmd_timeout = s->res_cmd_timeout;
  ra.poll_naps = 1;
  if (const char* e = std::getenv("ABD_RES_POLL_NAPS")) ra.poll_naps = std::max(0, std::atoi(e));
#ifdef ABD_STAMPS
  ra.dbg_launch = (int)c->resident_launches;
  ra.dbg_unit = u;
#endif
  ra.guard_timeout = s->res_guard_timeout;
  r.status_h[0] = 0;
  r.status_h[1] = 0;
  c->seq += 1.0;
  r.rounds += 1;
  resident_write(r, cp, c->seq);  // the first command is there before the kernel starts
  if (c->storage == ABD_STORE_F32)
    hipLaunchKernelGGL(abd_dense_resident_kernel<float>, dim3(s->res_blocks), dim3(ABD_BLOCK), s->res_lds, pp.st, ra);
  else
    hipLaunchKernelGGL(abd_dense_resident_kernel<double>, dim3(s->res_blocks), dim3(ABD_BLOCK), s->res_lds, pp.st, ra);
  HIP_TRY(hipGetLastError());
  r.live = true;
  c->resident_launches++;
  c->resident_commands++;
  return ABD_OK;
}

// end of the trajectory: the kernel leaves (the next launch of this unit uses the other mailbox, so a kernel that is
// still on its way out never sees a command that is not meant for it)
void resident_quit(abd_sampler* s, int u) {
  abd_ctx* c = s->c;
  abd_sampler::Resident& r = s->res[(size_t)u];
  if (!r.live) return;
  c->seq += 1.0;
  volatile unsigned long long* m = r.mail_h + (size_t)r.cur * ABD_RES_WORDS;
  const double q = c->seq + 0.5;
  unsigned long long sq;
  std::memcpy(&sq, &q, sizeof sq);
  __atomic_store_n(const_cast<unsigned long long*>(m) + 7, sq, __ATOMIC_RELEASE);
  __atomic_store_n(const_cast<unsigned long long*>(m) + 15, sq, __ATOMIC_RELEASE);
  r.cur ^= 1;
  r.live = false;
}

// the kernel of unit u left on its own (no command for cmd_timeout): start over from a clean counter
int resident_recover(abd_sampler* s, int u) {
  abd_ctx* c = s->c;
  abd_sampler::Resident& r = s->res[(size_t)u];
  HIP_TRY(hipStreamSynchronize(c->pipe[unit_pipe(c, u)].st));
  HIP_TRY(hipMemset(r.relay + ABD_RES_RELAY_WORDS, 0, sizeof(unsigned long long)));
  r.rounds = 0;
  r.cur ^= 1;
  r.live = false;
  c->resident_restarts++;
  return ABD_OK;
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
  if (const char* e = std::getenv("ABD_SAMPLER_THREADS")) s->threads = std::max(1, std::min(16, std::atoi(e)));
  // With four host threads (observation lists) the best split is four units -- one per thread and per
  // hardware queue: default cohort, evaluations/s seen by NUTS with 4 / 8 / 16 chains 217 k / 339 k / 491 k against
  // 166 k / 253 k / 300-370 k for the best split on one thread.
  // Large dense cohorts (one host thread): at most about four units -- the hardware queues -- of 1, 2, 4 or 8 chains, the
  // sizes the dense kernel has a shape for (config 3, evaluations/s seen by NUTS over 150-300 iterations: 8 chains 79 k
  // with units of 1, 112 k with 2; 12 chains 97 k / 88 k / 83 k with 2 / 3 / 4; 16 chains 91 k / 100 k with 2 / 4;
  // 32 chains 100 k / 134 k with 4 / 8)
  int dense_unit = 1;
  while (dense_unit < 8 && 2 * dense_unit <= n / 4) dense_unit *= 2;
  s->unit = (c->dense && (int64_t)c->G * c->N >= 500000) ? dense_unit : std::max(s->threads > 1 ? 1 : 2, std::min(8, (n + 3) / 4));
  if (const char* e = std::getenv("ABD_SAMPLER_UNIT")) s->unit = std::atoi(e);
  s->unit = std::max(1, std::min({s->unit, n, (int)ABD_MAX_BATCH}));

  // several units' launches are in flight: one workgroup per CU each, whatever the number of units -- a unit's numbers
  // must not depend on it
  c->group_blocks = std::min(c->dense_blocks, c->n_cu);
  if (const char* e = std::getenv("ABD_GROUP_BLOCKS_PER_CU")) c->group_blocks = std::max(1, std::min(c->n_cu * std::atoi(e), c->blocks_max));
  // the starting points through the launch shape the units will use
  rc = hipSetDevice(c->device) == hipSuccess ? flush_ring(c) : fail(ABD_ERR_HIP, "hipSetDevice failed");
  if (!rc && (n + s->unit - 1) / s->unit > 1 && !(std::getenv("ABD_PROBE_QUEUES") && std::atoi(std::getenv("ABD_PROBE_QUEUES")) == 0)) rc = probe_stream_queues(c);
  if (!rc) rc = resident_setup(s);
  for (int u = 0, lo = 0; lo < n && !rc; ++u, lo += s->unit) {
    const int m = std::min(s->unit, n - lo);
    rc = enqueue_slot(c, kSyncSlot + u, m, chains + lo, theta0 + (size_t)lo * ABD_N_THETA, true, false, unit_pipe(c, u));
    if (!rc) rc = wait_rows(c, kSyncSlot + u, m, c->seq, c->pipe[unit_pipe(c, u)].st);
    if (!rc) rc = fetch_slot(c, kSyncSlot + u, s->lp.data() + lo, s->gr.data() + (size_t)lo * ABD_N_THETA);
  }
  if (rc) {
    resident_free(s);
    delete s;
    return rc;
  }
  for (int k = 0; k < n; ++k) {
    if (!std::isfinite(s->lp[(size_t)k])) {
      resident_free(s);
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
      resident_free(s);
      delete s;
      return fail(ABD_ERR_HIP, "sampler sums: %s", hipGetErrorString(e));
    }
  }
  *out = s;
  return ABD_OK;
}

void abd_sampler_destroy(abd_sampler* s) {
  if (!s) return;
  if (!s->res.empty()) {
    (void)hipSetDevice(s->c->device);
    (void)hipDeviceSynchronize();
    resident_free(s);
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
int record_stage_chain(abd_sampler* s, const abd_record* rec, int k, int64_t pos, hipStream_t st) {
  abd_ctx* c = s->c;
  const size_t cells = (size_t)c->G * c->N, N = (size_t)c->N;
  const size_t per_var = (size_t)s->n * s->rec_chunk * cells;
  const size_t lds = (size_t)3 * (c->G + 1) * sizeof(double2_t);
  const int blocks = std::max(1, std::min((c->N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 8));
  const size_t at = ((size_t)k * s->rec_chunk + pos);
  const ChainSlot& slot = c->slots[(size_t)s->chains[(size_t)k]];
  if (rec->i || rec->ab_n_mu || rec->ab_s_mu) {
    EvalArgs a;
    base_args(c, a);
    a.n_chains = 1;
    a.ch[0] = chain_par(c, s->chains[(size_t)k], s->ch[(size_t)k].nuts.q);
    hipLaunchKernelGGL(abd_deterministics_kernel, dim3(blocks), dim3(ABD_BLOCK), lds, st, a,
                       rec->i ? s->d_rec_i8 + per_var + at * cells : (int8_t*)nullptr,
                       rec->ab_n_mu ? s->d_rec_mu + at * cells : (double*)nullptr,
                       rec->ab_s_mu ? s->d_rec_mu + per_var + at * cells : (double*)nullptr, (double*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  if (rec->i_raw) {
    dim3 grid((c->N + 255) / 256, c->G);
    hipLaunchKernelGGL(abd_unpack_bits_kernel, grid, dim3(256), 0, st, slot.rw, s->d_rec_i8 + at * cells, c->G, c->N);
    HIP_TRY(hipGetLastError());
  }
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
  // host threads (see below): a power of two <= 8, so that units that share a HIP stream (u and u + 8) share their thread
  int T_all = 1;
  while (2 * T_all <= std::min({s->threads, n_units, (int)kMaxPipes})) T_all *= 2;
  if (s->unit_seq.size() != (size_t)n_units) {
    s->unit_seq.resize((size_t)n_units);
    for (int u = 0; u < n_units; ++u) s->unit_seq[(size_t)u] = (double)(u + 1) * 1099511627776.0;  // (u + 1) 2^40
  }
  HIP_TRY(hipSetDevice(c->device));
  if (int frc = flush_ring(c)) return frc;
  HIP_TRY(hipStreamSynchronize(c->stream));  // whatever the caller queued on the context's stream comes first
  auto stream_of = [&](int u) { return c->pipe[unit_pipe(c, u)].st; };
  // evaluate the points th[0 .. m) of the unit's chains who[0 .. m)
  auto launch_eval = [&](int u) -> int {
    Unit& un = units[(size_t)u];
    double* seqp = T_all > 1 ? &s->unit_seq[(size_t)u] : nullptr;
    int rc = enqueue_slot(c, kSyncSlot + u, un.m, un.ids.data(), un.th.data(), true, false, unit_pipe(c, u), seqp);
    if (rc) return rc;
    un.tag = seqp ? *seqp : c->seq;
    un.t_queued = std::chrono::steady_clock::now();
    return ABD_OK;
  };
  auto launch_tree = [&](int u) -> int {  // the next leapfrog of every tree of the unit that is still growing
    Unit& un = units[(size_t)u];
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
    if (!s->resident) return launch_eval(u);
    // one chain per unit: its evaluation kernel stays on the device until the tree stops (abd_resident.hpp)
    if (int rc = resident_eval(s, u, un.ids[0], un.th.data())) return rc;
    un.tag = c->seq;
    return ABD_OK;
  };
  struct QuitGuard {  // whichever way this function is left, no resident kernel keeps waiting for commands
    abd_sampler* s;
    ~QuitGuard() {
      for (size_t u = 0; u < s->res.size(); ++u) resident_quit(s, (int)u);
    }
  } quit_guard{s};
  auto ready = [&](int u) -> bool {  // have all result rows of the unit's launch landed? (never blocks)
    const Unit& un = units[(size_t)u];
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
      }
      if (draw && s->d_sums)
        if (int rc = accumulate_chain(s, j, st)) return rc;
      if (recording)
        if (int rc = record_stage_chain(s, rec, j, un.staged, st)) return rc;
    }
    if (recording && ++un.staged == s->rec_chunk) {
      for (int j = un.lo; j < un.hi; ++j)
        if (int rc = record_flush_chain(s, rec, j, un.flushed_to, un.staged, st)) return rc;
      un.flushed_to += un.staged;
      un.staged = 0;
    }
    un.k += 1;
    if (un.k == n_iter) {
      un.state = DONE;
      if (recording)
        for (int j = un.lo; j < un.hi; ++j)
          if (int rc = record_flush_chain(s, rec, j, un.flushed_to, un.staged, st)) return rc;
      return ABD_OK;
    }
    for (int j = un.lo; j < un.hi; ++j) s->ch[(size_t)j].begin();
    un.state = EVAL;
    return launch_tree(u);
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
  static const bool profile = std::getenv("ABD_SAMPLER_PROFILE") != nullptr;
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
        if (!ready(u)) {
          if (!(s->resident && s->res[(size_t)u].live && *(volatile unsigned int*)s->res[(size_t)u].status_h == 2u)) continue;
          // the kernel gave up waiting (the host was away for longer than its time-out).  Once it has left for good,
          // either the answer has landed after all (it may leave while the last workgroup is still summing) or the
          // command was never seen: relaunch and ask again
          if (int rc = resident_recover(s, u)) return rc;
          if (!ready(u)) {
            if (++s->res[(size_t)u].fails > 50) return fail(ABD_ERR_STATE, "the resident evaluation kernel of chain %d does not answer", un.ids[0]);
            if (int rc = resident_eval(s, u, un.ids[0], un.th.data())) return rc;
            un.tag = c->seq;
            continue;
          }
        }
        if (profile && !progressed) t_handle = clk::now();
        if (profile) prof_wait += std::chrono::duration<double>(clk::now() - un.t_queued).count();
        progressed = true;
        ++handled;
        if (s->resident) s->res[(size_t)u].fails = 0;
        clk::time_point tp0;
        if (profile) tp0 = clk::now();
        if (int frc = fetch_slot(c, kSyncSlot + u, un.lp.data(), un.gr.data())) return frc;
        if (profile) {
          const clk::time_point t1 = clk::now();
          prof_fetch += std::chrono::duration<double>(t1 - tp0).count();
          tp0 = t1;
        }
        if (un.state == EVAL) {
          for (int q = 0; q < un.m; ++q)
            s->ch[(size_t)un.who[(size_t)q]].nuts.feed(un.lp[(size_t)q], un.gr.data() + (size_t)q * ABD_N_THETA);
          if (profile) {
            const clk::time_point t1 = clk::now();
            prof_feed += std::chrono::duration<double>(t1 - tp0).count();
            tp0 = t1;
          }
          const int lrc = launch_tree(u);
          if (profile) prof_launch += std::chrono::duration<double>(clk::now() - tp0).count();
          if (lrc) return lrc;
          if (un.m) continue;  // some tree of the unit is still growing
          if (s->resident) resident_quit(s, u);
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
      errors[(size_t)tid] = g_err;  // the message is thread-local: hand it to the calling thread
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

}  // namespace

int abd_sampler_run_record(abd_sampler* s, int64_t n_iter, double* theta, double* stats, const abd_record* rec) {
  if (!s) return fail(ABD_ERR_ARG, "sampler is NULL");
  if (n_iter < 0) return fail(ABD_ERR_ARG, "n_iter=%lld is negative", (long long)n_iter);
  abd_ctx* c = s->c;
  const int n = s->n;
  const bool recording = rec && (rec->i_raw || rec->ab_s_waner || rec->i || rec->ab_n_mu || rec->ab_s_mu);
  if (recording) {
    if (rec->first < 0 || rec->first + n_iter > rec->capacity)
      return fail(ABD_ERR_ARG, "record: draws [%lld, %lld) do not fit capacity %lld", (long long)rec->first,
                  (long long)(rec->first + n_iter), (long long)rec->capacity);
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

}  // extern "C"
