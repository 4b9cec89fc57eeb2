// abd_eval.hip -- evaluation launches of the C ABI (include/abd_hip.h): the dense-panel and observation-list kernels,
// the pipes (HIP streams) stream-ordered launches rotate over, completion tags in mapped host memory, device timing.
#include "abd_host.hpp"
#include "abd_eval_kernels.hpp"
#include "abd_train.hpp"

namespace abdi {

LaunchProfile g_launch_profile;

size_t table_lds_bytes(int G, int cpw, int red_rows, bool exp2_tab = false);
size_t dense_lds_bytes(int G, int cpw) { return abd_dense_lds(G, cpw); }
size_t table_lds_bytes(int G, int cpw, int red_rows, bool exp2_tab) {
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
hipError_t launch_dense_g(bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a, bool xc) {
  if (xc)  // split panels: R + 1 instead of 2 R bytes per cell and antigen
    return grad ? launch_k(abd_dense_kernel<R, C, true, true>, grid, lds, st, a) : launch_k(abd_dense_kernel<R, C, false, true>, grid, lds, st, a);
  return grad ? launch_k(abd_dense_kernel<R, C, true, false>, grid, lds, st, a) : launch_k(abd_dense_kernel<R, C, false, false>, grid, lds, st, a);
}
template <typename R>
hipError_t launch_dense(int C, bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a, bool xc) {
  switch (C) {
    case 4: return launch_dense_g<R, 4>(grad, grid, lds, st, a, xc);
    case 2: return launch_dense_g<R, 2>(grad, grid, lds, st, a, xc);
    default: return launch_dense_g<R, 1>(grad, grid, lds, st, a, xc);
  }
}
template <typename R, int C>
hipError_t launch_sparse_g(bool grad, dim3 grid, size_t lds, hipStream_t st, const EvalArgs& a) {
  if (a.nt > ABD_MAXT)  // more than 256 gaps: the kernels that keep an individual's words in registers are built for 8 words too
    return grad ? launch_k(abd_sparse_kernel<R, C, true, ABD_MAXT_MAX>, grid, lds, st, a)
                : launch_k(abd_sparse_kernel<R, C, false, ABD_MAXT_MAX>, grid, lds, st, a);
  return grad ? launch_k(abd_sparse_kernel<R, C, true, ABD_MAXT>, grid, lds, st, a)
              : launch_k(abd_sparse_kernel<R, C, false, ABD_MAXT>, grid, lds, st, a);
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
  if (a.nt > ABD_MAXT)
    return grad ? launch_k(abd_obs_kernel<R, true, ABD_MAXT_MAX>, grid, lds, st, a) : launch_k(abd_obs_kernel<R, false, ABD_MAXT_MAX>, grid, lds, st, a);
  return grad ? launch_k(abd_obs_kernel<R, true, ABD_MAXT>, grid, lds, st, a) : launch_k(abd_obs_kernel<R, false, ABD_MAXT>, grid, lds, st, a);
}

// does a dense launch with cpw chains per workgroup read the split panels (od + one-byte dilution code) or the pair panels?
bool dense_xc(const abd_ctx* c, int cpw) { return c->xc_ok && cpw <= c->xc_max_cb; }

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
int dense_blocks(const abd_ctx* c, int cpw, int share, int grid_rows) {
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
// fixed-order sum of an earlier launch -- are fin_rows shorter and the others share the difference.  The kernel works its
// range out from these five numbers (abd_types.hpp: EvalArgs::rg_*).
template <typename ARGS>
void range_split(const abd_ctx* c, int blocks, int nsub, ARGS& a, bool fused_sums = true) {
  const int64_t rows_total = (int64_t)c->n_lg * c->G, n_ranges = (int64_t)blocks * nsub;
  const int64_t n_short = (nsub == 1 && fused_sums) ? std::min<int64_t>(ABD_MAX_BATCH, n_ranges) : 0;
  const int64_t e_fin = (nsub == 1 && (rows_total + n_short * c->fin_rows) / n_ranges >= 2 * c->fin_rows) ? c->fin_rows : 0;
  const int64_t virt = rows_total + n_short * e_fin;
  a.rg_base = (int32_t)(virt / n_ranges);
  a.rg_extra = (int32_t)(virt % n_ranges);
  a.rg_e_fin = (int32_t)e_fin;
  a.rg_n_short = (int32_t)n_short;
  a.rg_g_magic = abd_div_magic((uint32_t)c->G);
}

// Which of the context's streams can have kernels on the device at the same time?  HIP multiplexes its streams over a few
// hardware queues (4 by default) and a queue runs one kernel after the other, whichever stream it came from.  One wave
// per stream that stays for 150 us, launched back to back: a stream whose wave starts only when an earlier stream's
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
                  bool deferred = false, int force_pipe = -1, const HostTerms* host = nullptr, double* seqp = nullptr,
                  TrainArgs* train = nullptr, bool sync_call = false) {
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
    lds = train ? abd_obs_lds_own_sum(c->G) : abd_obs_lds_head(c->G);
  } else if (c->dense) {
    // the last launch of a batch of stream-ordered steps ends alone on the chip: it gets the grid of a launch that has the
    // chip to itself (one wave per SIMD issues at half the rate; a K = 20 region 376 -> 373 us; a longer tail did not pay)
    blocks = dense_blocks(c, cpw, force_pipe >= 0 ? 2 : (rotate && c->steps_behind != 0 ? 1 : 0), n / cpw);
    lds = abd_dense_lds(c->G, cpw, dense_xc(c, cpw));
  } else {
    blocks = c->blocks_x;
    lds = table_lds_bytes(c->G, cpw, ABD_WAVES_PER_BLOCK * cpw);
  }
  if (blocks > c->blocks_max) return fail(ABD_ERR_STATE, "internal: grid %d exceeds partial rows %d", blocks, c->blocks_max);
  if (c->dense && !lanes)
    range_split(c, blocks, ABD_WAVES_PER_BLOCK / cpw, a);
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
  // a sampler unit's dense launch sums its own partial rows: no second launch
  if (train && pp.on)
    if (int frc = flush_pipe(c, pi)) return frc;  // a train launch sums its own rows: nothing may be pending on its pipe
  // (the observation-lane kernel sums its own rows only in a train launch: for an evaluation the host waits for, the second
  // launch is as fast -- profiles/README.md, history)
  // (a synchronous call's dense launch too: the host waits for nothing but its rows, and a second launch is 3.6 us of it)
  const bool sync_own = sync_call && c->sync_own_sum && c->dense && !lanes && !pp.on && c->timing == 0 &&
                        (int64_t)n * (blocks + ABD_TRAIN_SHARDS) <= (int64_t)c->n_slots * c->blocks_max;
  const bool fused_sum = (sync_own || (force_pipe >= 0 && c->dense_own_sum && !(c->fuse_finalize && pp.on))) && ((c->dense && !lanes) || (lanes && train));
  if (train) {
    if (!fused_sum || n != 1) return fail(ABD_ERR_STATE, "internal: a leapfrog-train launch needs one chain and a kernel that sums its own rows");
    train->dense = c->dense ? 1 : 0;
    train->tag = seq + 1.0;
    a.train = *train;
  }
  if (fused_sum) {
    if (c->dense && !lanes && blocks > ABD_TRAIN_ONE_LEVEL)  // a grid that fills the chip: two-level count-in (abd_dense.hpp)
      a.fin_count2 = c->d_train_count + (size_t)pi * ABD_MAX_BATCH * (1 + ABD_TRAIN_SHARDS) * ABD_TRAIN_CNT_STRIDE;
    else
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
    le = c->storage == ABD_STORE_F32 ? launch_dense<float>(cpw, grad, grid, lds, pp.st, a, dense_xc(c, cpw))
                                     : launch_dense<double>(cpw, grad, grid, lds, pp.st, a, dense_xc(c, cpw));
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
  return flush_pending(c);
}

// Wait for the rows of a synchronous call (written into mapped host memory) by polling their completion tag;
// falls back to a stream synchronise if it does not show up quickly.
int wait_rows(abd_ctx* c, int slot, int n, double tag, hipStream_t st) {
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
  // the tag did not show within ~2 M polls (tens of ms): the stream synchronise below is always correct, but it should
  // never be needed, so it is counted (abd_wait_fallbacks) instead of passing as a slow call -- and a row that still
  // lacks its tag afterwards is an error, not a result
  __atomic_fetch_add(&c->wait_fallbacks, (int64_t)1, __ATOMIC_RELAXED);
  HIP_TRY(hipStreamSynchronize(st ? st : c->stream));
  for (int q = first; q < n; ++q)
    if (rows[(size_t)q * ABD_NOUT + ABD_NOUT - 1] != tag) return fail(ABD_ERR_STATE, "result row %d of slot %d never received its completion tag", q, slot);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return ABD_OK;
}

// Wait until every row of a stream-ordered slot carries its completion tag (the rows are in mapped host memory).
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
  __atomic_fetch_add(&c->wait_fallbacks, (int64_t)1, __ATOMIC_RELAXED);  // counted, then verified: see wait_rows
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int q = 0; q < r.n; ++q)
    if (rows[(size_t)q * ABD_NOUT + ABD_NOUT - 1] != r.tag_first + (double)(q / ABD_MAX_BATCH))
      return fail(ABD_ERR_STATE, "result row %d of stream-ordered slot %d never received its completion tag", q, slot);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return ABD_OK;
}

int enqueue_slot(abd_ctx* c, int slot, int n, const int32_t* chains, const double* theta, bool grad, bool deferred,
                 int force_pipe, double* seqp, bool sync_call) {
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
  // every result row lives in mapped host memory (one PCIe write of 16 doubles + tag per chain, ~3 us inside the kernel)
  double* rows = c->d_out + (size_t)slot * c->n_slots * ABD_NOUT;
  r.tag_first = (seqp ? *seqp : c->seq) + 1.0;
  if (deferred) c->pending_slots.push_back(slot);
  for (int k0 = 0; k0 < n; k0 += ABD_MAX_BATCH) {
    const int m = std::min(ABD_MAX_BATCH, n - k0);
    rc = enqueue_group(c, m, chains + k0, theta + (size_t)k0 * ABD_N_THETA, grad, rows + (size_t)k0 * ABD_NOUT, deferred, force_pipe,
                       r.host.data() + k0, seqp, nullptr, sync_call);
    if (!rc && force_pipe >= 0) rc = flush_pipe(c, force_pipe);  // a group's fixed-order sum follows on its own stream
    if (rc) return rc;
  }
  return ABD_OK;
}

// One launch of a leapfrog train (abd_types.hpp: TrainArgs; abd_sampler.hip) for `chain` on pipe `pi`: the launch assembles
// its own result and leaves it in t->rec under t->tag (set here); nothing is kept in the result slots.
int enqueue_train_launch(abd_ctx* c, int chain, int pi, TrainArgs* t, const HostTerms& first_terms, double* seqp) {
  const int32_t ch = chain;
  return enqueue_group(c, 1, &ch, nullptr, true, c->d_out + (size_t)kSyncSlot * c->n_slots * ABD_NOUT, false, pi, &first_terms, seqp, t);
}

// ---- leapfrog-train launches of dense cohorts (abd_train.hpp; abd_sampler.hip) ----
template <typename R, int CB>
hipError_t launch_train_cb(bool xc, dim3 grid, size_t lds, hipStream_t st, const DenseTrainArgs& a) {
  auto go = [&](auto kernel) -> hipError_t {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(ABD_BLOCK), lds, st, a);
    return hipGetLastError();
  };
  return xc ? go(abd_train_kernel<R, CB, true>) : go(abd_train_kernel<R, CB, false>);
}
template <typename R>
hipError_t launch_train(int cb, bool xc, dim3 grid, size_t lds, hipStream_t st, const DenseTrainArgs& a) {
  switch (cb) {
    case 4: return launch_train_cb<R, 4>(xc, grid, lds, st, a);
    case 2: return launch_train_cb<R, 2>(xc, grid, lds, st, a);
    default: return launch_train_cb<R, 1>(xc, grid, lds, st, a);
  }
}

// Queue one launch of a train unit of cb (1, 2 or 4) chains on pipe pi: a->tc[0 .. cb) filled by the caller, everything
// else here.  `blocks`: workgroups with a range (the unit's fixed shape: its chains' numbers depend on nothing else).
int enqueue_dense_train(abd_ctx* c, int pi, int cb, int blocks, DenseTrainArgs* a) {
  if (!c->dense) return fail(ABD_ERR_STATE, "internal: dense train launch on a cohort kept as observation lists");
  if (cb != 1 && cb != 2 && cb != 4) return fail(ABD_ERR_STATE, "internal: train unit of %d chains", cb);
  bool any_step = false, any_fwd = false;
  for (int k = 0; k < cb; ++k) {
    TrainChainArgs& tc = a->tc[k];
    if (tc.action != ABD_TR_SKIP || tc.fwd_slot >= 0) {
      if (!tc.st || !tc.ring || !tc.iw || !tc.cnt || !tc.waner || (tc.action == ABD_TR_BEGIN && !tc.begin))
        return fail(ABD_ERR_STATE, "internal: train launch with a NULL pointer for chain %d of the unit", k);
      if ((tc.action == ABD_TR_STEP && (tc.use_slot | 1) != 1) || tc.fwd_slot > 1)
        return fail(ABD_ERR_STATE, "internal: train launch with slot %d / %d for chain %d of the unit", tc.use_slot, tc.fwd_slot, k);
    }
    any_step |= tc.action == ABD_TR_STEP;
    any_fwd |= tc.fwd_slot >= 0;
  }
  for (int k = cb; k < ABD_TRAIN_CB; ++k) {
    std::memset(&a->tc[k], 0, sizeof a->tc[k]);
    a->tc[k].fwd_slot = -1;
  }
  if (!any_step) blocks = 1;  // nothing to walk: one workgroup runs the state machines (new transitions)
  if (blocks < 1 || blocks > c->blocks_max) return fail(ABD_ERR_STATE, "internal: train grid %d outside [1, %d]", blocks, c->blocks_max);
  const bool xc = c->xc_ok;
  a->yx_n = c->n.yx;
  a->yx_s = c->s.yx;
  a->od_n = c->n.od;
  a->od_s = c->s.od;
  a->xc_n = c->n.xc;
  a->xc_s = c->s.xc;
  a->dict_n = c->n.dict;
  a->dict_s = c->s.dict;
  a->n_dict_n = c->n.n_dict;
  a->n_dict_s = c->s.n_dict;
  a->vw = c->vw;
  a->exp2_tab = c->exp2_tab;
  abd_ctx::Pipe& pp = c->pipe[pi];
  if (pp.on)
    if (int frc = flush_pipe(c, pi)) return frc;  // (a pending fixed-order sum of an earlier plain launch on this stream)
  if (pi > 0) pp.busy = true;
  a->partials = pp.partials[0];
  a->fin_count = c->d_train_count + (size_t)pi * ABD_MAX_BATCH * (1 + ABD_TRAIN_SHARDS) * ABD_TRAIN_CNT_STRIDE;
  if ((int64_t)cb * (blocks + ABD_TRAIN_SHARDS) > (int64_t)c->n_slots * c->blocks_max)
    return fail(ABD_ERR_STATE, "internal: train launch of %d x %d workgroups exceeds the partial rows", cb, blocks);
  a->prior_const = c->prior_const;
  a->xcd_remap = c->xcd_remap ? 1 : 0;
  a->service = any_fwd ? 1 : 0;
  // the service workgroup takes a workgroup slot: a grid that fills the chip exactly leaves it one (the ranges are equal
  // shares of the plane whatever their number), or the last range would only start when the first workgroup has left.
  // ALWAYS, whether this launch carries a service workgroup or not: the split of the plane decides the order of the sums, and
  // a chain's numbers must not depend on what its unit's other chains happen to have pending
  if (any_step && blocks > 1 && blocks % c->n_cu == 0) blocks -= 1;
  a->G = c->G;
  a->N = c->N;
  a->n_lg = c->n_lg;
  a->K_n = (int32_t)c->n.K;
  a->K_s = (int32_t)c->s.K;
#ifdef ABD_STAMPS
  a->stamps = stamps_buffer();
#endif
  range_split(c, blocks, ABD_WAVES_PER_BLOCK / cb, *a, false);
  const size_t lds = abd_dense_lds(c->G, cb, xc, true);
  dim3 grid(blocks + a->service, 1);
  std::chrono::steady_clock::time_point lp0;
  if (g_launch_profile.on) lp0 = std::chrono::steady_clock::now();
  const hipError_t le = c->storage == ABD_STORE_F32 ? launch_train<float>(cb, xc, grid, lds, pp.st, *a) : launch_train<double>(cb, xc, grid, lds, pp.st, *a);
  if (g_launch_profile.on) {
    g_launch_profile.eval_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - lp0).count();
    g_launch_profile.evals++;
  }
  HIP_TRY(le);
  return ABD_OK;
}

int fetch_slot(abd_ctx* c, int slot, double* logp, double* grad, bool with_priors) {
  if (slot < 0 || slot >= kSyncSlot + c->n_sync_slots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slot, kResultSlots);
  const ResultSlot& r = c->results[slot];
  if (r.n == 0) return fail(ABD_ERR_STATE, "result slot %d is empty", slot);
  const double* rows = c->h_out + (size_t)slot * c->n_slots * ABD_NOUT;
  for (int k = 0; k < r.n; ++k)
    assemble(c, r.host[(size_t)k], r.theta.data() + (size_t)k * ABD_N_THETA, rows + (size_t)k * ABD_NOUT, logp + k,
             (grad && r.grad) ? grad + (size_t)k * ABD_N_THETA : nullptr, with_priors);
  return ABD_OK;
}

}  // namespace abdi

extern "C" {

int abd_n_result_slots(abd_ctx*) { return kResultSlots; }

int abd_logp_dlogp_batch_enqueue(abd_ctx* c, int32_t slot, int32_t n, const int32_t* chains, const double* theta) {
  if (!c || !chains || !theta) return fail(ABD_ERR_ARG, "NULL argument");
  if (slot < 0 || slot >= kResultSlots) return fail(ABD_ERR_ARG, "result slot %d outside [0, %d)", slot, kResultSlots);
  return enqueue_slot(c, slot, n, chains, theta, true, true);
}

int abd_wait(abd_ctx* c) {
  if (!c) return fail(ABD_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  int rc = flush_ring(c);
  if (rc) return rc;
  // every stream-ordered slot's rows carry a tag: the newest slots land last, so poll backwards and stop early
  for (size_t q = c->pending_slots.size(); q-- > 0;)
    if (int wrc = wait_slot(c, c->pending_slots[q])) return wrc;
  c->pending_slots.clear();
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
    for (int k = s0; k < s1; ++k) {
      c->steps_behind = n_steps - 1 - k;
      const int rc = enqueue_slot(c, k - s0, n, chains, theta + (size_t)k * per_step, grad != nullptr, true);
      c->steps_behind = -1;
      if (rc) return rc;
    }
    // the results land in mapped host memory slot by slot: queue the pending sums and the joins, then take every step's
    // result as soon as its tag shows -- the host-side assembly of the early steps overlaps the late steps' kernels
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = flush_ring(c)) return rc;
    for (int k = s0; k < s1; ++k) {
      if (int rc = wait_slot(c, k - s0)) return rc;
      if (int rc = fetch_slot(c, k - s0, logp + (size_t)k * n, grad ? grad + (size_t)k * per_step : nullptr)) return rc;
    }
    c->pending_slots.clear();
  }
  return ABD_OK;
}

int abd_logp_dlogp_batch(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta, double* logp, double* grad) {
  if (!c || !chains || !theta || !logp || !grad) return fail(ABD_ERR_ARG, "NULL argument");
  if (int frc = flush_ring(c)) return frc;
  int rc = enqueue_slot(c, kSyncSlot, n, chains, theta, true, false, -1, nullptr, true);
  if (rc) return rc;
  if (int prc = flush_pending(c)) return prc;  // (a launch that did not sum its own rows: the sum follows right away)
  if (int wrc = wait_rows(c, kSyncSlot, c->results[kSyncSlot].n, c->seq)) return wrc;
  return fetch_slot(c, kSyncSlot, logp, grad);
}

int abd_logp_dlogp(abd_ctx* c, int32_t chain, const double* theta, double* logp, double* grad) {
  return abd_logp_dlogp_batch(c, 1, &chain, theta, logp, grad);
}

int abd_loglik_dlogp(abd_ctx* c, int32_t chain, const double* theta, double* loglik, double* grad) {
  if (!c || !theta || !loglik || !grad) return fail(ABD_ERR_ARG, "NULL argument");
  int rc = enqueue_slot(c, kSyncSlot, 1, &chain, theta, true, false, -1, nullptr, true);
  if (rc) return rc;
  if (int prc = flush_pending(c)) return prc;
  if (int wrc = wait_rows(c, kSyncSlot, c->results[kSyncSlot].n, c->seq)) return wrc;
  return fetch_slot(c, kSyncSlot, loglik, grad, false);
}

int abd_logp(abd_ctx* c, int32_t chain, const double* theta, double* logp) {
  if (!c || !theta || !logp) return fail(ABD_ERR_ARG, "NULL argument");
  int rc = enqueue_slot(c, kSyncSlot, 1, &chain, theta, false, false, -1, nullptr, true);
  if (rc) return rc;
  if (int prc = flush_pending(c)) return prc;
  if (int wrc = wait_rows(c, kSyncSlot, c->results[kSyncSlot].n, c->seq)) return wrc;
  return fetch_slot(c, kSyncSlot, logp, nullptr);
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

}  // extern "C"
