// abd_sparse.hpp -- observation lists, one WAVE per individual (persistent, grid-stride), 1 or 2 chains per wave sharing
// the loads; masks on the scalar unit.  For lists so full that a wave stays busy for two or more rounds; otherwise the
// lane-per-observation kernel (abd_obs.hpp) is used.
#pragma once

#include "abd_device.hpp"

template <typename R, int CPW, bool GRAD, int MT>
__global__ __launch_bounds__(ABD_BLOCK) void abd_sparse_kernel(const EvalArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  // LDS: [CPW][2][G+1] power tables + [G+1] "ones" table + block reduction
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  const int G = a.G;
  const int N = a.N;
  const int nt = a.nt;
  const int tstride = G + 1;
  double2_t* tab_ones = tabs + CPW * 2 * tstride;
  double* red = reinterpret_cast<double*>(tab_ones + tstride);  // [WAVES][CPW][ABD_NOUT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cbase = blockIdx.y * CPW;

#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    fill_pow_table(tabs + (c * 2 + 0) * tstride, a.ch[cbase + c].rho_n, tstride, tid, ABD_BLOCK);
    fill_pow_table(tabs + (c * 2 + 1) * tstride, a.ch[cbase + c].rho_s, tstride, tid, ABD_BLOCK);
  }
  fill_ones_table(tab_ones, tstride, tid, ABD_BLOCK);
  __syncthreads();

  double acc[CPW][ABD_NACC];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int k = 0; k < ABD_NACC; ++k) acc[c][k] = 0.0;
  // sum(i_raw), sum(ab_s_waner): the slot's counters, carried into the sums by the chain's first wave
  double n1[CPW], m1[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const bool first = blockIdx.x == 0 && wave == 0;
    n1[c] = first ? (double)a.ch[cbase + c].cnt[0] : 0.0;
    m1[c] = first ? (double)a.ch[cbase + c].cnt[1] : 0.0;
  }

  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    // the individual's vaccination words and each chain's CONSTRAINED infection words (kept with the chain slot:
    // abd_small.hpp: abd_constrain_kernel), wave-uniform
    uint64_t V[MT], I[CPW][MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      V[t] = 0;
#pragma unroll
      for (int c = 0; c < CPW; ++c) I[c][t] = 0;
      if (t < nt) {
        V[t] = uniform_word(a.vw, (int64_t)t * N + j);
#pragma unroll
        for (int c = 0; c < CPW; ++c) I[c][t] = uniform_word(a.ch[cbase + c].iw, (int64_t)t * N + j);
      }
    }
    int wj[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      wj[c] = __builtin_amdgcn_readfirstlane((int)a.ch[cbase + c].waner[j]) != 0;
    }

#pragma unroll
    for (int ag = 0; ag < 2; ++ag) {
      const int32_t* ptr = ag == 0 ? a.ptr_n : a.ptr_s;
      const uint16_t* gi = ag == 0 ? a.g_n : a.g_s;
      const void* yy = ag == 0 ? a.y_n : a.y_s;
      const void* xx = ag == 0 ? a.x_n : a.x_s;
      const int k0 = ptr[j], k1 = ptr[j + 1];
      for (int kb = k0; kb < k1; kb += 64) {
        const int k = kb + lane;
        const bool in = k < k1;
        const int kk = in ? k : k0;
        const int g = gi[kk];
        const double y = ld<R>(yy, kk), x = ld<R>(xx, kk);
        const double guard = in ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const ChainPar& p = a.ch[cbase + c];
          const double2_t* tn = tabs + (c * 2 + 0) * tstride;
          const double2_t* ts = wj[c] ? tabs + (c * 2 + 1) * tstride : tab_ones;
          const Resp rs = responses<MT>(g, nt, I[c], V, tn, ts);
          double h = 0.0;
          if (ag == 0) {
            const double an = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
            obs_term<GRAD>(an, x, y, p.b_n, p.d_n, guard, acc[c][A_N_Q2], acc[c][A_N_H], acc[c][A_N_HX], acc[c][A_N_QS], h);
            if (GRAD) {
              acc[c][A_N_HC] += rs.cum_i ? h : 0.0;
              acc[c][A_N_HU] = fma(h, rs.un, acc[c][A_N_HU]);
              acc[c][A_N_HD] = fma(h, rs.dn, acc[c][A_N_HD]);
            }
          } else {
            const double as = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
            obs_term<GRAD>(as, x, y, p.b_s, p.d_s, guard, acc[c][A_S_Q2], acc[c][A_S_H], acc[c][A_S_HX], acc[c][A_S_QS], h);
            if (GRAD) {
              acc[c][A_S_HC] += rs.cum_iv ? h : 0.0;
              acc[c][A_S_HD] = fma(h, rs.ds, acc[c][A_S_HD]);
            }
          }
        }
      }
    }
  }

  // ---- reduction: lanes -> wave -> block (LDS) -> per-block partial in global memory ----
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
#pragma unroll
    for (int k = 0; k < ABD_NACC; ++k) {
      const double v = wave_sum(acc[c][k]);
      if (lane == 0) red[(wave * CPW + c) * ABD_NOUT + k] = v;
    }
    if (lane == 0) {
      red[(wave * CPW + c) * ABD_NOUT + ABD_NACC] = n1[c];
      red[(wave * CPW + c) * ABD_NOUT + ABD_NACC + 1] = m1[c];
      red[(wave * CPW + c) * ABD_NOUT + ABD_NACC + 2] = 0.0;
    }
  }
  __syncthreads();
  if (tid < CPW * ABD_NOUT) {
    const int c = tid / ABD_NOUT, k = tid % ABD_NOUT;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < ABD_WAVES_PER_BLOCK; ++w) v += red[(w * CPW + c) * ABD_NOUT + k];
    a.partials[((int64_t)(cbase + c) * gridDim.x + blockIdx.x) * ABD_NOUT + k] = v;
  }
}
