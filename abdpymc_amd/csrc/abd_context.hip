// abd_context.hip -- context life cycle, discrete state and host-side closed forms of the C ABI (include/abd_hip.h).
//
// Replaces, for the joint-logp path only, what PyMC/PyTensor compile out of abdpymc.model()
// (reference abdpymc/abd.py:396-469): the closed-form prior terms + transform Jacobians are evaluated
// here on the host (17 scalars), the O(G*N) data term on the device (abd_dense.hpp, abd_obs.hpp, abd_sparse.hpp).
#include "abd_host.hpp"

#include <unordered_map>
#include "abd_small.hpp"

namespace abdi {

namespace {
thread_local std::string g_err = "";
}
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
const char* last_error() { return g_err.c_str(); }
void set_error(const std::string& msg) { g_err = msg; }

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

ModelSizes model_sizes(const abd_ctx* c) {
  ModelSizes m;
  m.G = c->G;
  m.dense = c->dense ? 1 : 0;
  m.N = (double)c->N;
  m.cells = (double)c->G * (double)c->N;
  m.Kn = (double)c->n.K;
  m.Ks = (double)c->s.K;
  m.prior_const = c->prior_const;
  return m;
}

// Combine the device sums of one chain with the closed-form terms (abd_terms.hpp)
void assemble(const abd_ctx* c, const HostTerms& h, const double* t, const double* sums, double* logp, double* grad,
              bool with_priors) {
  assemble_terms(model_sizes(c), h, t, sums, logp, grad, with_priors);
}

ChainPar chain_par(const abd_ctx* c, int chain, const Transformed& tr) {
  ChainPar p;
  chain_par_from_tr(p, &tr.p);
  p.rw = c->slots[chain].rw;
  p.waner = c->slots[chain].waner;
  p.iw = c->slots[chain].iw;
  p.cnt = c->slots[chain].cnt;
  return p;
}
ChainPar chain_par(const abd_ctx* c, int chain, const double* t) { return chain_par(c, chain, transform(t)); }

ConstrainArgs constrain_args(const abd_ctx* c) {
  ConstrainArgs a;
  std::memset(&a, 0, sizeof a);
  a.pw = c->ignore_pcr ? nullptr : c->pw;
  a.N = c->N;
  a.nt = c->nt;
  a.n_chunks = c->n_chunks;
  std::memcpy(a.chunk_mask, c->chunk_mask, sizeof a.chunk_mask);
  return a;
}

#ifdef ABD_STAMPS
unsigned long long* stamps_buffer() {
  static unsigned long long* stamps = nullptr;
  if (!stamps) {
    (void)hipHostMalloc((void**)&stamps, 4096 * 16 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
    if (const char* e = std::getenv("ABD_STAMPS_PTR_OUT")) {  // the probe reads the buffer through its address
      FILE* f = std::fopen(e, "w");
      if (f) {
        std::fprintf(f, "%llu\n", (unsigned long long)(uintptr_t)stamps);
        std::fclose(f);
      }
    }
  }
  return stamps;
}
#endif

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
  a.yxi_n = c->n.yxi;
  a.yxi_s = c->s.yxi;
  a.od_n = c->n.od;
  a.od_s = c->s.od;
  a.xc_n = c->n.xc;
  a.xc_s = c->s.xc;
  a.dict_n = c->n.dict;
  a.dict_s = c->s.dict;
  a.n_dict_n = c->n.n_dict;
  a.n_dict_s = c->s.n_dict;
  a.vw = c->vw;
  a.pw = c->ignore_pcr ? nullptr : c->pw;
  a.exp2_tab = c->exp2_tab;
#ifdef ABD_STAMPS
  a.stamps = stamps_buffer();
#endif
  a.G = c->G;
  a.N = c->N;
  a.nt = c->nt;
  a.n_chunks = c->n_chunks;
  a.n_lg = c->n_lg;
  std::memcpy(a.chunk_mask, c->chunk_mask, sizeof a.chunk_mask);
}
int probe_stream_queues(abd_ctx* c) {
  if (c->n_queues > 0) return ABD_OK;
  HIP_TRY(hipSetDevice(c->device));
  // Candidates: the context's streams plus as many spare ones.  HIP hands its hardware queues to streams in creation order
  // across the whole process, so the context's 8 streams rarely sit on them evenly (3 + 2 + 2 + 1 on this stack): after the
  // probe, streams of over-subscribed queues are exchanged for spare streams of under-subscribed ones, so that eight units
  // are two per queue, not three on one (the queue with the most units sets the pace of a run).
  constexpr int kCand = 2 * kMaxPipes;
  const int ns = c->n_streams;
  hipStream_t cand[kCand] = {};
  int n_cand = ns;
  for (int pi = 0; pi < ns; ++pi) cand[pi] = c->pipe[pi].st;
  for (; n_cand < ns + kMaxPipes; ++n_cand)
    if (hipStreamCreateWithFlags(&cand[n_cand], hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
  unsigned long long* d = nullptr;
  hipError_t le = hipMalloc(&d, (size_t)kCand * 2 * sizeof(unsigned long long));
  // Stream k's wave stays for 150 + 20 k us, so the waves end at least 20 us apart and a wave that had to wait for a
  // queue starts within a few us of exactly one earlier wave's end: it is behind that one.  A wave that starts while all
  // earlier ones are still there has a queue to itself.  A host hiccup between two launches can make a stream look
  // queued, never the other way round: up to three attempts, the one that finds the most queues counts.
  int best_nq = 0, best[kCand] = {};
  for (int attempt = 0; attempt < 3 && best_nq < 4 && le == hipSuccess; ++attempt) {
    le = hipDeviceSynchronize();
    for (int k = 0; k < n_cand && le == hipSuccess; ++k) {
      hipLaunchKernelGGL(abd_spin_kernel, dim3(1), dim3(64), 0, cand[k], d + 2 * k, 15000ull + 2000ull * (unsigned long long)k);
      le = hipGetLastError();
    }
    if (le == hipSuccess) le = hipDeviceSynchronize();
    unsigned long long h[kCand * 2] = {};
    if (le == hipSuccess) le = hipMemcpy(h, d, sizeof(unsigned long long) * 2 * (size_t)n_cand, hipMemcpyDeviceToHost);
    if (le != hipSuccess) break;
    int nq = 0, q_of[kCand] = {};
    unsigned long long busy_until[kCand] = {};
    for (int j = 0; j < n_cand; ++j) {
      int q = -1;
      for (int k = 0; k < nq && q < 0; ++k)
        if (h[2 * j] + 300 >= busy_until[k] && h[2 * j] <= busy_until[k] + 800) q = k;  // started 0-8 us after queue k drained
      if (q < 0) q = nq++;
      q_of[j] = q;
      busy_until[q] = h[2 * j + 1];
    }
    if (nq > best_nq) {
      best_nq = nq;
      std::copy(q_of, q_of + kCand, best);
    }
  }
  if (d) (void)hipFree(d);
  const int nq = std::max(1, best_nq);
  // exchange: pipe 0 is the context's main stream and stays; a pipe whose queue holds more than its share gives its stream
  // up for a spare one on the queue that holds the fewest
  if (le == hipSuccess && best_nq > 1) {
    int load[kCand] = {};
    bool used[kCand] = {};
    for (int pi = 0; pi < ns; ++pi) load[best[pi]]++;
    for (int pi = ns - 1; pi >= 1; --pi) {
      const int q = best[pi];
      int q_min = 0;
      for (int k = 1; k < nq; ++k)
        if (load[k] < load[q_min]) q_min = k;
      if (load[q] - load[q_min] < 2) continue;
      int spare = -1;
      for (int k = ns; k < n_cand && spare < 0; ++k)
        if (!used[k] && best[k] == q_min) spare = k;
      if (spare < 0) continue;
      used[spare] = true;
      std::swap(cand[pi], cand[spare]);
      c->pipe[pi].st = cand[pi];
      best[pi] = q_min;
      load[q]--;
      load[q_min]++;
    }
  }
  for (int k = ns; k < n_cand; ++k)
    if (cand[k]) (void)hipStreamDestroy(cand[k]);
  HIP_TRY(le);
  // queue numbers in order of first appearance among the context's streams
  int renum[kCand], n_seen = 0;
  std::fill(renum, renum + kCand, -1);
  for (int pi = 0; pi < ns; ++pi) {
    if (renum[best[pi]] < 0) renum[best[pi]] = n_seen++;
    c->queue_of_pipe[pi] = renum[best[pi]];
  }
  c->n_queues = std::max(1, n_seen);
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
int check_chains(abd_ctx* c, int n, const int32_t* chains) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  if (n < 1 || n > c->n_slots) return fail(ABD_ERR_ARG, "n=%d outside [1, n_chain_slots=%d]", n, c->n_slots);
  for (int k = 0; k < n; ++k) {
    if (chains[k] < 0 || chains[k] >= c->n_slots) return fail(ABD_ERR_ARG, "chain %d outside [0, %d)", chains[k], c->n_slots);
    if (!c->slots[chains[k]].set) return fail(ABD_ERR_STATE, "chain slot %d has no discrete state (call abd_set_discrete)", chains[k]);
  }
  return ABD_OK;
}

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
    {  // the individual-major copy the sweep kernels read: element (g, j) at [j * G + g]
      std::vector<YX<R>> yxi((size_t)G * N);
      for (int j = 0; j < N; ++j)
        for (int g = 0; g < G; ++g) yxi[(size_t)j * G + g] = yx[(size_t)g * N + j];
      HIP_TRY(hipMalloc(&d.yxi, yxi.size() * sizeof(YX<R>)));
      HIP_TRY(hipMemcpy(d.yxi, yxi.data(), yxi.size() * sizeof(YX<R>), hipMemcpyHostToDevice));
    }
    // the split panels of one-chain launches: od alone + one byte per cell coding its log dilution (assays use a handful of
    // dilutions; lossless: the dictionary holds the values of the pair panel, i.e. rounded to the storage type).  Lane-group-major: element (g, j) at
    // [((j / 64) * G + g) * 64 + j % 64], so the rows a wave walks -- 64 individuals, gap after gap -- are one contiguous
    // stream (a 64-byte code row is half a cache line: in gap-major order its other half belongs to the neighbouring lane
    // group and was fetched again by that group's wave, 1.29 x the algorithmic bytes at config 5)
    std::vector<double> dict;
    std::unordered_map<uint64_t, uint8_t> code_of;  // bit pattern of a log dilution -> its code
    const size_t n_lg = ((size_t)N + 63) / 64;
    auto cell_of = [&](size_t g, size_t j) { return ((j / 64) * (size_t)G + g) * 64 + j % 64; };
    std::vector<uint8_t> code(n_lg * (size_t)G * 64, 0);
    bool fits = true;
    for (size_t k = 0; k < K && fits; ++k) {
      const double x = (double)(R)o.log_dilution[k];  // what the pair panel holds: the value in the storage type
      uint64_t bits;
      std::memcpy(&bits, &x, sizeof bits);  // bit-wise: -0.0, NaN payloads stay what they are
      auto it = code_of.find(bits);
      if (it == code_of.end()) {
        if (dict.size() == ABD_XDICT) {
          fits = false;
          break;
        }
        it = code_of.emplace(bits, (uint8_t)dict.size()).first;
        dict.push_back(x);
      }
      code[cell_of((size_t)o.idx_gap[k], (size_t)o.idx_ind[k])] = it->second;
    }
    if (fits) {
      std::vector<R> od(code.size(), (R)0);
      for (size_t j = 0; j < (size_t)N; ++j)
        for (size_t g = 0; g < (size_t)G; ++g) od[cell_of(g, j)] = yx[g * N + j].y;
      d.n_dict = (int)dict.size();
      HIP_TRY(hipMalloc(&d.od, od.size() * sizeof(R)));
      HIP_TRY(hipMemcpy(d.od, od.data(), od.size() * sizeof(R), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&d.xc, code.size()));
      HIP_TRY(hipMemcpy(d.xc, code.data(), code.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&d.dict, ABD_XDICT * sizeof(double)));
      dict.resize(ABD_XDICT, 0.0);
      HIP_TRY(hipMemcpy(d.dict, dict.data(), ABD_XDICT * sizeof(double), hipMemcpyHostToDevice));
    }
    return ABD_OK;
  }
  std::vector<R> y(std::max<size_t>(K, 1)), x(std::max<size_t>(K, 1));
  std::vector<uint16_t> g(std::max<size_t>(K, 1));
  std::vector<int32_t> jj(std::max<size_t>(K, 1));
  for (size_t k = 0; k < K; ++k) {
    const int64_t src = so.order[k];
    y[k] = (R)o.od[src];
    x[k] = (R)o.log_dilution[src];
    g[k] = (uint16_t)o.idx_gap[src];
    jj[k] = (int32_t)o.idx_ind[src];
  }
  HIP_TRY(hipMalloc(&d.j, jj.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d.j, jj.data(), jj.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&d.y, y.size() * sizeof(R)));
  HIP_TRY(hipMalloc(&d.x, x.size() * sizeof(R)));
  HIP_TRY(hipMemcpy(d.y, y.data(), y.size() * sizeof(R), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.x, x.data(), x.size() * sizeof(R), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&d.g, g.size() * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(d.g, g.data(), g.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
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
    if (a->od) (void)hipFree(a->od);
    if (a->xc) (void)hipFree(a->xc);
    if (a->yxi) (void)hipFree(a->yxi);
    if (a->dict) (void)hipFree(a->dict);
  }
  if (c->vw) (void)hipFree(c->vw);
  if (c->pw) (void)hipFree(c->pw);
  if (c->exp2_tab) (void)hipFree(c->exp2_tab);
  if (c->stage_gn) (void)hipFree(c->stage_gn);
  for (auto& s : c->slots) {
    if (s.rw) (void)hipFree(s.rw);
    if (s.waner) (void)hipFree(s.waner);
    if (s.iw) (void)hipFree(s.iw);
    if (s.cnt) (void)hipFree(s.cnt);
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
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->d_work) (void)hipFree(c->d_work);
  if (c->d_counts_chain) (void)hipFree(c->d_counts_chain);
  if (c->d_fin_count) (void)hipFree(c->d_fin_count);
  if (c->d_train_count) (void)hipFree(c->d_train_count);
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

int launch_deterministics(abd_ctx* c, int chain, const double* theta, hipStream_t st, int8_t* out_i, double* out_mun, double* out_mus,
                          double* sums) {
  EvalArgs a;
  base_args(c, a);
  a.n_chains = 1;
  a.ch[0] = chain_par(c, chain, theta);
  const size_t lds = (size_t)3 * (c->G + 1) * sizeof(double2_t);
  const int blocks = std::max(1, std::min((c->N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 8));
  if (c->nt > ABD_MAXT)
    hipLaunchKernelGGL(abd_deterministics_kernel<ABD_MAXT_MAX>, dim3(blocks), dim3(ABD_BLOCK), lds, st, a, out_i, out_mun, out_mus, sums);
  else
    hipLaunchKernelGGL(abd_deterministics_kernel<ABD_MAXT>, dim3(blocks), dim3(ABD_BLOCK), lds, st, a, out_i, out_mun, out_mus, sums);
  HIP_TRY(hipGetLastError());
  return ABD_OK;
}

int launch_unpack(abd_ctx* c, int chain, int8_t* dst, hipStream_t st) {
  dim3 grid((c->N + 255) / 256, c->G);
  hipLaunchKernelGGL(abd_unpack_bits_kernel, grid, dim3(256), 0, st, c->slots[(size_t)chain].rw, dst, c->G, c->N);
  HIP_TRY(hipGetLastError());
  return ABD_OK;
}

}  // namespace abdi

extern "C" {

const char* abd_version(void) { return "abdpymc_amd hip gfx950 0.2"; }

const char* abd_last_error(void) { return last_error(); }

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
  // the dense kernel addresses the gap rows of a piece (up to G of them) with a 32-bit scalar offset (abd_dense.hpp);
  // beyond 2^28 cells (fp64; 2^29 in fp32 storage) per GPU the cohort takes the observation-list kernels instead
  if ((int64_t)N * (d->storage == ABD_STORE_F32 ? 8 : 16) * (G + 2) >= ((int64_t)1 << 32)) c->dense = false;
  // ... and splits the (lane group, gap) plane by 32-bit arithmetic: row / G by a 32-bit reciprocal must be exact for every
  // row of the plane (abd_types.hpp: abd_div_magic_exact; 6.8 M individuals at 200 gaps -- beyond the limit above anyway)
  if (!abd_div_magic_exact((uint64_t)c->n_lg * (uint64_t)G, (uint32_t)G)) c->dense = false;
  if (env_int("ABD_FORCE_SPARSE", 0)) c->dense = false;
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
  const int bpc = std::max(1, tune_int("ABD_BLOCKS_PER_CU", 2));
  const int sparse_max = std::max(1, std::min((N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 16));
  c->blocks_x = std::max(1, std::min(sparse_max, c->n_cu * bpc));
  // dense kernel: 4 workgroups per CU = 4 waves per SIMD (<= 128 VGPRs, ~29 KB LDS each): one round, equal ranges
  const int dbpc = std::max(1, tune_int("ABD_DENSE_BLOCKS_PER_CU", 4));
  {
    const int cap = c->n_cu * 8;
    c->ob_n = (int)std::min<int64_t>((d->n.n_obs + ABD_BLOCK - 1) / ABD_BLOCK, cap);
    c->ob_s = (int)std::min<int64_t>((d->s.n_obs + ABD_BLOCK - 1) / ABD_BLOCK, cap);
    c->ob_c = 1;  // one workgroup carries the slot's counters (sum(i_raw), sum(ab_s_waner)) into the sums
    // lane per observation unless the lists are so full that a wave per individual keeps its 64 lanes busy
    // for two rounds or more and amortises the constraint pass (measured crossover, tools/bench_sparse.py)
    c->obs_lanes = d->s.n_obs + d->n.n_obs < (int64_t)256 * N;
    c->obs_lanes = env_int("ABD_OBS_LANES", c->obs_lanes ? 1 : 0) != 0;
  }
  c->blocks_max = std::max({sparse_max, c->n_cu * 16, c->ob_n + c->ob_s + c->ob_c});
  c->dense_blocks = std::min(c->n_cu * dbpc, c->blocks_max);
  if (dense_lds_bytes(G, 4) > 160 * 1024) {
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
  c->xc_ok = c->dense && c->s.od && c->n.od;
  c->xc_max_cb = tune_int("ABD_XC_MAX_CB", c->xc_max_cb);
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
    CREATE_TRY(hipMalloc(&s.iw, words * sizeof(uint64_t)));
    CREATE_TRY(hipMalloc(&s.cnt, 2 * sizeof(long long)));
  }
  c->pipe[0].st = c->stream;
  c->n_pipes = std::max(1, std::min(6, env_int("ABD_PIPES", c->n_pipes)));
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
  if (const int pb = tune_int("ABD_PIPE_BLOCKS", 0)) c->pipe_blocks = std::max(1, std::min(pb, c->blocks_max));
  c->dbpc = dbpc;
  c->group_blocks = std::min(c->dense_blocks, c->n_cu * std::max(1, dbpc / 2));
  for (int pi = 0; pi < kMaxPipes; ++pi)
    if (c->pipe[pi].st)
      for (int b = 0; b < 2; ++b)
        CREATE_TRY(hipMalloc(&c->pipe[pi].partials[b], (size_t)c->n_slots * c->blocks_max * ABD_NOUT * sizeof(double)));
  c->fuse_finalize = tune_int("ABD_FUSE_FINALIZE", 1) != 0;
  c->xcd_remap = tune_int("ABD_XCD_REMAP", 1) != 0;
  c->fin_rows = std::max(0, tune_int("ABD_FIN_ROWS", 2));
  const size_t out_bytes = (size_t)(kResultSlots + c->n_sync_slots) * c->n_slots * ABD_NOUT * sizeof(double);
  // COHERENT (fine-grained) on purpose: synchronous calls poll a completion tag in this memory while the stream
  // is still running.  With hipHostMallocMapped alone the allocation is non-coherent: the GPU caches it and the
  // two 64-byte halves of a result row could reach the host in either order (tag visible, data stale).
  CREATE_TRY(hipHostMalloc(&c->h_out, out_bytes, hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(c->h_out, 0, out_bytes);
  CREATE_TRY(hipHostGetDevicePointer((void**)&c->d_out, c->h_out, 0));
  CREATE_TRY(hipMalloc(&c->d_counts, ((size_t)c->n_slots * 2 + 8) * sizeof(unsigned long long)));  // + 8 development counters
  CREATE_TRY(hipMalloc(&c->d_work, (size_t)2 * c->n_slots * sizeof(unsigned int)));
  CREATE_TRY(hipMalloc(&c->d_counts_chain, (size_t)c->n_slots * 2 * sizeof(unsigned long long)));
  CREATE_TRY(hipMalloc(&c->d_fin_count, (size_t)kMaxPipes * ABD_MAX_BATCH * sizeof(unsigned int)));
  CREATE_TRY(hipMemset(c->d_fin_count, 0, (size_t)kMaxPipes * ABD_MAX_BATCH * sizeof(unsigned int)));
  {
    const size_t tc_bytes = (size_t)kMaxPipes * ABD_MAX_BATCH * (1 + ABD_TRAIN_SHARDS) * ABD_TRAIN_CNT_STRIDE * sizeof(unsigned int);
    CREATE_TRY(hipMalloc(&c->d_train_count, tc_bytes));
    CREATE_TRY(hipMemset(c->d_train_count, 0, tc_bytes));
  }
  c->dense_own_sum = env_int("ABD_DENSE_OWN_SUM", 1) != 0;
  // measured (round 4, profiles/r04: sync_own_sum_ab.txt): 28.8 / 33.8 / 43.1 us per call of 1 / 2 / 4 chains at config 3 with the
  // launch summing its own rows (two-level count-in at 1 024 workgroups), 28.1 / 32.0 / 41.9 us with the sum as a second
  // launch: the second launch stays
  c->sync_own_sum = c->dense_own_sum && tune_int("ABD_SYNC_OWN_SUM", 0) != 0;
  CREATE_TRY(hipHostMalloc(&c->h_counts_chain, (size_t)c->n_slots * 2 * sizeof(unsigned long long), hipHostMallocDefault));
  c->gibbs_v1 = env_int("ABD_GIBBS_V1", 0) != 0;
  c->g2_refill_min = std::max(1, std::min(64, tune_int("ABD_G2_REFILL_MIN", ABD_G2_REFILL_MIN)));
  c->g2_tail_lanes = std::max(0, std::min(64, tune_int("ABD_G2_TAIL_LANES", ABD_G2_TAIL_LANES)));
  c->g2_tail_age = std::max(0, tune_int("ABD_G2_TAIL_AGE", ABD_G2_TAIL_AGE));
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
  // what the slot keeps beside the raw state: constrained words, sum(i_raw), sum(ab_s_waner)
  HIP_TRY(hipMemsetAsync(s.cnt, 0, 2 * sizeof(long long), c->stream));
  if (c->nt > ABD_MAXT)
    hipLaunchKernelGGL(abd_constrain_kernel<ABD_MAXT_MAX>, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, constrain_args(c), s.rw,
                       s.waner, s.iw, reinterpret_cast<unsigned long long*>(s.cnt));
  else
    hipLaunchKernelGGL(abd_constrain_kernel<ABD_MAXT>, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, constrain_args(c), s.rw,
                       s.waner, s.iw, reinterpret_cast<unsigned long long*>(s.cnt));
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
  if (c->nt > ABD_MAXT)
    hipLaunchKernelGGL(abd_flip_kernel<ABD_MAXT_MAX>, dim3(1), dim3(1), 0, c->stream, constrain_args(c), s.rw, s.waner, s.iw,
                       reinterpret_cast<unsigned long long*>(s.cnt), c->G, flat);
  else
    hipLaunchKernelGGL(abd_flip_kernel<ABD_MAXT>, dim3(1), dim3(1), 0, c->stream, constrain_args(c), s.rw, s.waner, s.iw,
                       reinterpret_cast<unsigned long long*>(s.cnt), c->G, flat);
  HIP_TRY(hipGetLastError());
  return ABD_OK;
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
  if (int lrc = launch_deterministics(c, chain, theta, c->stream, i ? d_i : (int8_t*)nullptr, mu_n ? d_n : (double*)nullptr,
                                      mu_s ? d_s : (double*)nullptr, nullptr))
    return lrc;
  if (mu_n) HIP_TRY(hipMemcpyAsync(mu_n, d_n, cells * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (mu_s) HIP_TRY(hipMemcpyAsync(mu_s, d_s, cells * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (i) HIP_TRY(hipMemcpyAsync(i, d_i, cells, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ABD_OK;
}

// Queue one sweep launch for m <= ABD_MAX_BATCH chains on stream st (nothing is waited for): counts of chain k of the
int abd_get_discrete(abd_ctx* c, int32_t chain, int8_t* i_raw, int8_t* waner) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  int rc = check_chains(c, 1, &chain);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if (int jrc = join_pipes(c)) return jrc;
  ChainSlot& s = c->slots[(size_t)chain];
  if (i_raw) {
    if (int urc = launch_unpack(c, chain, c->stage_gn, c->stream)) return urc;
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
int abd_stream_queues(abd_ctx* c, int32_t* queue_of_stream, int32_t n) {
  if (!c || !queue_of_stream) return fail(ABD_ERR_ARG, "NULL argument");
  if (int rc = probe_stream_queues(c)) return rc;
  for (int i = 0; i < n; ++i) queue_of_stream[i] = i < c->n_streams ? c->queue_of_pipe[i] : -1;
  return ABD_OK;
}

int64_t abd_algorithmic_bytes(abd_ctx* c, int32_t n_chains) {
  if (!c) return 0;
  const int64_t R = c->storage == ABD_STORE_F32 ? 4 : 8;
  const int64_t cells = (int64_t)c->G * c->N;
  // indicator panels are bit-packed in 64-gap words: vacs + pcrpos + one i_raw per chain, plus the waner bytes
  const int64_t bits = (int64_t)c->nt * c->N * 8 * (2 + n_chains) + (int64_t)n_chains * c->N;
  // one chain per launch reads the split panels (od + a one-byte dilution code per cell and antigen) where they exist
  const int cpw = n_chains % 4 == 0 ? 4 : (n_chains % 2 == 0 ? 2 : 1);  // abd_eval.hip: pick_cpw
  if (c->dense && c->xc_ok && cpw <= c->xc_max_cb) return cells * 2 * (R + 1) + bits;
  if (c->dense) return cells * 4 * R + bits;
  return (c->s.K + c->n.K) * (2 * R + 2) + 2 * (int64_t)(c->N + 1) * 4 + bits;
}

}  // extern "C"
