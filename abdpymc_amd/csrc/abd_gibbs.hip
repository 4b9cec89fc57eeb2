// abd_gibbs.hip -- the device Gibbs sweep of the C ABI (abd_gibbs_sweep; include/abd_hip.h): what PyMC's
// BinaryGibbsMetropolis does to [i_raw, ab_s_waner] inside pm.sample (abd.py:922).
#include "abd_host.hpp"
#include "abd_gibbs.hpp"
#include "abd_gibbs2.hpp"

namespace abdi {

// the wave-per-proposal sweep kernel for 4 or 8 words per individual (<= 256 / <= 512 gaps); observation lists of at most 64
// gaps -- the reference's own cohorts have 26 and 31 -- get a one-word instantiation: this kernel holds the individual's packed
// rows in scalar registers, and with one word instead of four they fit (no spills; profiles/r04)
template <typename R, bool DENSE>
void launch_gibbs_v1(int nt, dim3 grid, size_t lds, hipStream_t st, const GibbsArgs& ga) {
  if constexpr (!DENSE) {
    if (nt == 1) {
      hipLaunchKernelGGL((abd_gibbs_kernel<R, DENSE, 1>), grid, dim3(ABD_BLOCK), lds, st, ga);
      return;
    }
  }
  if (nt > ABD_MAXT)
    hipLaunchKernelGGL((abd_gibbs_kernel<R, DENSE, ABD_MAXT_MAX>), grid, dim3(ABD_BLOCK), lds, st, ga);
  else
    hipLaunchKernelGGL((abd_gibbs_kernel<R, DENSE, ABD_MAXT>), grid, dim3(ABD_BLOCK), lds, st, ga);
}

int enqueue_gibbs(abd_ctx* c, int m, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
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
  int nw2 = abd_g2_waves(c->G, rbytes);  // waves of a workgroup = of a CU: as many as its LDS holds, 12 at most
  // a one-chain sweep is a sampler unit's: with 8 waves per CU (2 per SIMD) the other units' evaluation kernels find room
  // beside it (12 waves x 168 registers fill the CU's register file); the sweep itself 0.39 -> 0.41 ms, the compound
  // iteration of 4 chains at config 3 6.17 -> 5.90 ms.  Trajectories do not depend on the launch shape.
  if (m == 1) nw2 = std::max(4, std::min(nw2, tune_int("ABD_G2_WAVES_ONE", 8)));
  if (c->dense && !c->gibbs_v1 && nw2 >= 4) {  // (4 or 8 words per individual: <= 256 / <= 512 gaps)
    // lanes = proposals (abd_gibbs2.hpp): one workgroup per CU, the individuals of a chain handed out from one queue
    // per chain
    const size_t lds2 = abd_g2_lds(c->G, rbytes, nw2);
    HIP_TRY(hipMemsetAsync(work_dev, 0, (size_t)m * sizeof(unsigned int), st));
    const int bx = std::max(1, std::min(c->n_cu / m, (c->N + nw2 - 1) / nw2));
    dim3 grid2(bx, m);
    auto launch2 = [&](auto kernel) -> hipError_t {
      if (lds2 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(kernel, grid2, dim3(64 * nw2), lds2, st, ga);
      return hipSuccess;
    };
    const bool f32 = c->storage == ABD_STORE_F32;
    if (c->nt > ABD_MAXT) {
      if (stats_dev)  // ABD_GIBBS_STATS=1: the variant with the scheduler's development counters
        HIP_TRY(f32 ? launch2(abd_gibbs_dense_kernel<float, true, ABD_MAXT_MAX>) : launch2(abd_gibbs_dense_kernel<double, true, ABD_MAXT_MAX>));
      else
        HIP_TRY(f32 ? launch2(abd_gibbs_dense_kernel<float, false, ABD_MAXT_MAX>) : launch2(abd_gibbs_dense_kernel<double, false, ABD_MAXT_MAX>));
    } else if (stats_dev) {
      HIP_TRY(f32 ? launch2(abd_gibbs_dense_kernel<float, true, ABD_MAXT>) : launch2(abd_gibbs_dense_kernel<double, true, ABD_MAXT>));
    } else {
      HIP_TRY(f32 ? launch2(abd_gibbs_dense_kernel<float, false, ABD_MAXT>) : launch2(abd_gibbs_dense_kernel<double, false, ABD_MAXT>));
    }
  } else {
    const int blocks = std::max(1, std::min((c->N + ABD_WAVES_PER_BLOCK - 1) / ABD_WAVES_PER_BLOCK, c->n_cu * 8));
    const size_t lds = (size_t)3 * (c->G + 1) * sizeof(double2_t) + (size_t)ABD_WAVES_PER_BLOCK * abd_gibbs_wave_lds(c->G);
    dim3 grid(blocks, m);
    if (c->dense) {
      if (c->storage == ABD_STORE_F32)
        launch_gibbs_v1<float, true>(c->nt, grid, lds, st, ga);
      else
        launch_gibbs_v1<double, true>(c->nt, grid, lds, st, ga);
    } else {
      if (c->storage == ABD_STORE_F32)
        launch_gibbs_v1<float, false>(c->nt, grid, lds, st, ga);
      else
        launch_gibbs_v1<double, false>(c->nt, grid, lds, st, ga);
    }
  }
  HIP_TRY(hipGetLastError());
  return ABD_OK;
}

}  // namespace abdi

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
  static const bool want_stats = env_int("ABD_GIBBS_STATS", 0) != 0;
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

extern "C" {

int abd_gibbs_sweep(abd_ctx* c, int32_t n, const int32_t* chains, const double* theta, uint64_t seed, uint32_t sweep,
                    int64_t* accepted, int64_t* proposed) {
  return gibbs_sweep_impl(c, n, chains, theta, seed, sweep, 0u, accepted, proposed);
}

}  // extern "C"
