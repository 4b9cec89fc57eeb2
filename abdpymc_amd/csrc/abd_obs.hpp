// abd_obs.hpp -- the sparse observation lists (the reference's real cohorts: several dilutions per serum
// sample, most (gap, individual) cells empty), one LANE per observation.
//
// The default cohort has 35 709 observations for 1 520 individuals (SURVEY config 1): a wave per individual
// leaves most lanes idle and serialises the chains of a launch inside one wave.  Here every observation of
// every chain is its own lane: it reads its individual's packed indicator words, applies the infection
// constraints to them in registers, evaluates the response at its own gap from the set bits (LDS power
// tables) and adds one logistic term.  Work per lane is independent, so a 4-chain launch on the default
// cohort is ~2 300 wavefronts: one round on the chip.
//
// grid.x = ob_n workgroups over the N-antigen list, then ob_s over the S-antigen list, then ob_c over the
// individuals (the Bernoulli counts n1 = sum i_raw, m1 = sum waner); grid.y = chain of the launch.
// Each workgroup writes one row of partials[chain][block][16] (zeros where its segment has no term); the
// fixed-order sum over rows (finalize_chain) is shared with the other kernels.
// Included by abd_eval_kernels.hpp (after abd_dense.hpp, whose leapfrog-train epilogue it shares).
#pragma once

#include "abd_dense.hpp"

// exposures of the set bits of `m` (word t) at or before gap g: sum of tab[g - pos + 1]
__device__ __forceinline__ void add_bits(uint64_t m, int t, int g, const double2_t* tab, double& u, double& d) {
  const int rel = g - t * 64;
  const uint64_t le = rel >= 63 ? ~0ull : (rel < 0 ? 0ull : ((2ull << rel) - 1ull));
  m &= le;
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    const double2_t pw = tab[rel - b + 1];
    u += pw.x;
    d += pw.y;
  }
}

__device__ __forceinline__ bool any_bits(uint64_t m, int t, int g) {
  const int rel = g - t * 64;
  const uint64_t le = rel >= 63 ? ~0ull : (rel < 0 ? 0ull : ((2ull << rel) - 1ull));
  return (m & le) != 0;
}

// LDS of abd_obs_kernel: two tables of G + 1 entries, [waves][8] wave sums, a flag
__host__ __device__ inline size_t abd_obs_lds_head(int G) {
  return (size_t)2 * (size_t)(G + 1) * sizeof(double2_t) + (size_t)ABD_WAVES_PER_BLOCK * 8 * sizeof(double) + 16;
}
// ... and of a launch that sums its own partial rows (a leapfrog-train launch: the last workgroup's fixed-order sum reuses
// the tables' space, ABD_FIN_PARTS rows of 16 doubles)
__host__ __device__ inline size_t abd_obs_lds_own_sum(int G) {
  const size_t head = abd_obs_lds_head(G), sum = (size_t)ABD_FIN_PARTS * ABD_NOUT * sizeof(double);
  return head > sum ? head : sum;
}

template <typename R, bool GRAD, int MT>
__global__ __launch_bounds__(ABD_BLOCK) void abd_obs_kernel(const EvalArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int G = a.G, N = a.N, nt = a.nt, tstride = G + 1;
  double2_t* tab = reinterpret_cast<double2_t*>(smem);  // this segment's power table
  double2_t* tab_ones = tab + tstride;                  // rho_j = 1 for individuals whose S response does not wane
  double* red = reinterpret_cast<double*>(tab_ones + tstride);  // [waves][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ChainPar p = a.ch[blockIdx.y];
  const int b = blockIdx.x;
  // leapfrog trains (abd_types.hpp: TrainArgs; one chain per launch): the point was left in device memory by the launch
  // before this one on the stream, and that launch's record is passed on to the host
  if (GRAD && a.train.enabled && a.train.use_slot >= 0) {
    abdi::chain_par_from_tr(p, a.train.slots[a.train.use_slot].tr);
    if (a.train.fwd_rec && b == 0 && wave == ABD_WAVES_PER_BLOCK - 1) train_forward_record(a.train, lane);
  }
  const int seg = b < a.ob_n ? 0 : (b < a.ob_n + a.ob_s ? 1 : 2);
  double acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = 0.0;

  if (seg == 0) {
    // ---- N antigen: a = init + [any infection so far] perm + temp * sum rho^(g - r)   (abd.py:330-343)
    fill_pow_table(tab, p.rho_n, tstride, tid, ABD_BLOCK);
    __syncthreads();
    for (int64_t k = (int64_t)b * ABD_BLOCK + tid; k < a.K_n; k += (int64_t)a.ob_n * ABD_BLOCK) {
      const int j = a.j_n[k];
      const int g = a.g_n[k];
      const double y = ld<R>(a.y_n, k), x = ld<R>(a.x_n, k);
      uint64_t I[MT];  // the individual's constrained infections (kept with the chain slot)
#pragma unroll
      for (int t = 0; t < MT; ++t) I[t] = t < nt ? p.iw[(int64_t)t * N + j] : 0ull;
      double un = 0.0, dn = 0.0;
      bool cum = false;
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if (t < nt) {
          cum |= any_bits(I[t], t, g);
          add_bits(I[t], t, g, tab, un, dn);
        }
      const double an = p.init_n + (cum ? p.perm_n : 0.0) + p.temp_n * un;
      double h = 0.0;
      obs_term<GRAD, false>(an, x, y, p.b_n, p.d_n, 1.0, acc[0], acc[1], acc[2], acc[3], h);
      if (GRAD) {
        acc[4] += cum ? h : 0.0;
        acc[5] = fma(h, un, acc[5]);
        acc[6] = fma(h, dn, acc[6]);
      }
    }
  } else if (seg == 1) {
    // ---- S antigen: a = init + [any infection or dose so far] perm + sum rho_j^(g - r), unit boosts (Q1)
    fill_pow_table(tab, p.rho_s, tstride, tid, ABD_BLOCK);
    fill_ones_table(tab_ones, tstride, tid, ABD_BLOCK);
    __syncthreads();
    const int b0 = b - a.ob_n;
    for (int64_t k = (int64_t)b0 * ABD_BLOCK + tid; k < a.K_s; k += (int64_t)a.ob_s * ABD_BLOCK) {
      const int j = a.j_s[k];
      const int g = a.g_s[k];
      const double y = ld<R>(a.y_s, k), x = ld<R>(a.x_s, k);
      uint64_t I[MT], V[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        I[t] = V[t] = 0;
        if (t < nt) {
          I[t] = p.iw[(int64_t)t * N + j];
          V[t] = a.vw[(int64_t)t * N + j];
        }
      }
      const bool wj = p.waner[j] != 0;
      const double2_t* ts = wj ? tab : tab_ones;
      double us = 0.0, ds = 0.0;
      bool cum = false;
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if (t < nt) {
          cum |= any_bits(I[t] | V[t], t, g);
          add_bits(I[t], t, g, ts, us, ds);  // an infection and a dose in the same gap both count (Q5)
          add_bits(V[t], t, g, ts, us, ds);
        }
      const double as = p.init_s + (cum ? p.perm_s : 0.0) + us;
      double h = 0.0;
      obs_term<GRAD, false>(as, x, y, p.b_s, p.d_s, 1.0, acc[0], acc[1], acc[2], acc[3], h);
      if (GRAD) {
        acc[4] += cum ? h : 0.0;
        acc[6] = fma(h, ds, acc[6]);
      }
    }
  } else {
    // ---- Bernoulli counts on the RAW indicators (Q2) and on waner: the slot's counters (abd_constrain_kernel keeps them)
    const int b0 = b - a.ob_n - a.ob_s;
    if (b0 == 0 && tid == 0) {
      acc[0] = (double)p.cnt[0];
      acc[1] = (double)p.cnt[1];
    }
  }

#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wave * 8 + k] = v;
  }
  __syncthreads();
  if (tid < ABD_NOUT) {
    // where each of the segment's sums goes in the 16-entry row
    int src = -1;
    if (seg == 0) {
      src = tid == A_N_Q2 ? 0 : tid == A_N_H ? 1 : tid == A_N_HX ? 2 : tid == A_N_QS ? 3 : tid == A_N_HC ? 4 : tid == A_N_HU ? 5 : tid == A_N_HD ? 6 : -1;
    } else if (seg == 1) {
      src = tid == A_S_Q2 ? 0 : tid == A_S_H ? 1 : tid == A_S_HX ? 2 : tid == A_S_QS ? 3 : tid == A_S_HC ? 4 : tid == A_S_HD ? 6 : -1;
    } else {
      src = tid == ABD_NACC ? 0 : tid == ABD_NACC + 1 ? 1 : -1;
    }
    double v = 0.0;
    if (src >= 0) {
#pragma unroll
      for (int w = 0; w < ABD_WAVES_PER_BLOCK; ++w) v += red[w * 8 + src];
    }
    double* row = a.partials + ((int64_t)blockIdx.y * gridDim.x + b) * ABD_NOUT;
    if (a.fin_count)
      __hip_atomic_store(row + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another workgroup of THIS launch
    else
      row[tid] = v;  // the fixed-order sum follows as its own launch (abd_finalize_kernel)
  }
  if (!a.fin_count) return;

  // ---- own fixed-order sum (a leapfrog-train launch: the host is not there to queue a second kernel): the workgroup that
  // counts in last sums the chain's partial rows itself.  Hand-off as in abd_dense_kernel: the row was stored write-through
  // by 16 lanes of wave 0, wave 0 drains its stores, one lane counts in with a returning agent-scope add, the last
  // workgroup re-reads the rows with device-coherent loads behind a barrier.
  int* flag = reinterpret_cast<int*>(red + ABD_WAVES_PER_BLOCK * 8);
  __syncthreads();
  if (wave == 0) {
    handoff_drain_stores();
    if (lane == 0) {
      const unsigned int old = handoff_count_in(a.fin_count + blockIdx.y);
      flag[0] = old + 1u == gridDim.x ? 1 : 0;
    }
  }
  __syncthreads();
  const bool last = flag[0] != 0;  // workgroup-uniform; read before the sum's scratch overwrites the flag
  __syncthreads();
  if (!last) return;
  handoff_acquire();
  const double* rows = a.partials + (int64_t)blockIdx.y * gridDim.x * ABD_NOUT;
  if (GRAD && a.train.enabled) {
    sum_chain_coherent<ABD_BLOCK>(rows, (int)gridDim.x, reinterpret_cast<double*>(smem), tid);
    if (wave == 0) train_epilogue(a, reinterpret_cast<double*>(smem), lane);
  } else {
    finalize_chain_coherent<ABD_BLOCK>(rows, (int)gridDim.x, a.fin_out + (int64_t)blockIdx.y * ABD_NOUT, reinterpret_cast<double*>(smem), tid,
                                       a.fin_tag);
  }
  if (tid == 0) __hip_atomic_store(a.fin_count + blockIdx.y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
