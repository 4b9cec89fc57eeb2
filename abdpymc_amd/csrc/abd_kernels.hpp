// abd_kernels.hpp -- gfx950 (CDNA4) device code of the abdpymc joint-logp hot path.
//
// What the reference computes per evaluation (abdpymc/abd.py) and how it is mapped here:
//
//   i = constrain(i_raw, pcrpos)          abd.py:640-667, 560-601, 732-818   -> wave-uniform 64-bit masks
//   perm_response / temp responses         abd.py:242-306                     -> closed form over set bits
//   mu[idx_gap, idx_ind] -> logistic -> N  abd.py:343, 393, 445-469, 556-557  -> per-lane fp64, register sums
//
// One 64-lane wavefront owns one individual at a time; the lanes run along the gap axis (the panels are
// held individual-major, (N, G), so a wave's loads are contiguous).  The individual's three indicator
// rows (i_raw, pcrpos, vacs) are turned into 64-bit masks with one v_cmp per 64 gaps (ballot), so the
// whole integer pre-pass -- one-infection-per-chunk, PCR+ precedence, the 3-gap refractory recurrence on
// its own output -- runs on the scalar unit on <= 4 words.
//
// The reference's dense decay design, out[c] = sum_r (rho^max(0,c-r) - [c<r]) e[r]  (abd.py:258-274),
// is evaluated literally as a sum over the (few) set bits r <= c of the exposure mask, with rho^k and
// k rho^(k-1) read from a per-block LDS table -- no (G,G,N) tensor and no sequential scan.
//
// Sums over individuals x gaps stay in registers for the life of the wave (grid-stride over
// individuals), are reduced once per wave, once per block through LDS, written as per-block partials
// and summed in a fixed order by a second tiny kernel: no float atomics, bit-reproducible.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ABD_MAXT 4          // 64-gap tiles per individual (G <= 256)
#define ABD_NACC 13         // floating sums per chain (see enum below)
#define ABD_NOUT 16         // ABD_NACC + n1 + m1, padded
#define ABD_MAX_BATCH_K 16  // chains per launch
#define ABD_WAVES_PER_BLOCK 4
#define ABD_BLOCK (64 * ABD_WAVES_PER_BLOCK)

// raw sums accumulated on the device; constants (-b, perm, rho(1-rho) ...) are applied on the host
enum {
  A_N_R2 = 0,   // sum r^2                       (N likelihood)
  A_N_H,        // sum h            -> init_n    h = (r/sigma) d s (1-s);  d ll/d a = -b h
  A_N_HC,       // sum h [cumI>0]   -> perm_n
  A_N_HU,       // sum h U_n        -> temp_n    U_n = sum_r rho_n^(g-r)
  A_N_HD,       // sum h dU_n/drho  -> rho_n
  A_N_HX,       // sum h (a - x)    -> b_n
  A_N_WS,       // sum (r/sigma) s  -> d_n
  A_S_R2,
  A_S_H,
  A_S_HC,
  A_S_HD,
  A_S_HX,
  A_S_WS,
};

struct ChainPar {
  double perm_n, temp_n, rho_n, init_n;
  double perm_s, rho_s, init_s;
  double b_n, d_n, isig_n;
  double b_s, d_s, isig_s;
  const int8_t* iraw;   // (N, G) individual-major device copy of the chain's i_raw
  const int8_t* waner;  // (N,)
};

struct EvalArgs {
  const void* y_n;  // od, antigen N            R[K_n]   (dense: K = N*G, k = j*G + g)
  const void* x_n;  // log_dilution, antigen N
  const void* y_s;
  const void* x_s;
  const uint8_t* g_n;  // sparse only: gap index of each observation
  const uint8_t* g_s;
  const int32_t* ptr_n;  // sparse only: CSR row pointers by individual, (N+1)
  const int32_t* ptr_s;
  const int8_t* vacs;  // (N, G)
  const int8_t* pcr;   // (N, G) or nullptr (ignore_pcrpos)
  double* partials;    // [n_chains][n_blocks_x][ABD_NOUT]
  int32_t G, N, nt, n_chunks;
  int32_t n_chains, pad_;
  uint64_t chunk_mask[3][ABD_MAXT];
  ChainPar ch[ABD_MAX_BATCH_K];
};

struct double2_t {
  double x, y;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Fill one power table: tab[0] = {0,0} (index for "exposure is in the future"), tab[k+1] = {rho^k, k rho^(k-1)}.
__device__ __forceinline__ void fill_pow_table(double2_t* tab, double rho, int G, int tid, int nthreads) {
  for (int e = tid; e <= G; e += nthreads) {
    double2_t v;
    if (e == 0) {
      v.x = 0.0;
      v.y = 0.0;
    } else {
      const int k = e - 1;
      // rho^(k-1) by binary powering (k <= 255 -> <= 8 squarings)
      double base = rho, acc = 1.0;
      int n = k > 0 ? k - 1 : 0;
      while (n) {
        if (n & 1) acc *= base;
        base *= base;
        n >>= 1;
      }
      if (k == 0) {
        v.x = 1.0;  // rho^0 = 1 also for rho = 0 (abd.py:258: rho**design with design = 0)
        v.y = 0.0;
      } else {
        v.x = acc * rho;
        v.y = (double)k * acc;
      }
    }
    tab[e] = v;
  }
}

// Wave-uniform integer pre-pass for one individual and one chain (abd.py:640-667).
//   raw/pcr : masks of i_raw / pcrpos, bit b of word t <-> gap 64 t + b
//   out     : the Deterministic "i"
__device__ __forceinline__ void constrain_masks(const uint64_t raw[ABD_MAXT], const uint64_t pcr[ABD_MAXT],
                                                const EvalArgs& a, uint64_t out[ABD_MAXT]) {
  uint64_t i0[ABD_MAXT];
  if (a.n_chunks <= 1) {
    // OneTimeChunk: where(i_raw + pcrpos > 0, 1, 0)   abd.py:643-647
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) i0[t] = raw[t] | pcr[t];
  } else {
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) i0[t] = 0;
    for (int c = 0; c < a.n_chunks; ++c) {
      // mask_multiple_infections on the chunk: keep the first 1   abd.py:818
      // incorporate_pcrpos: any PCR+ in the chunk replaces the whole chunk column   abd.py:771
      bool has_pcr = false;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) has_pcr |= (pcr[t] & a.chunk_mask[c][t]) != 0;
      bool found = false;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) {
        const uint64_t cm = a.chunk_mask[c][t];
        uint64_t r = raw[t] & cm;
        uint64_t first = found ? 0ull : (r & (0ull - r));
        found |= r != 0;
        i0[t] |= has_pcr ? (pcr[t] & cm) : first;
      }
    }
  }
  // mask_three_gaps: out[t] = in[t] unless out[t-1] | out[t-2] | out[t-3]   abd.py:560-601.
  // Greedy over set bits in ascending order is the same recurrence: a set bit is kept iff no kept bit
  // lies in the three gaps before it.
  int block_until = 0;
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    uint64_t m = i0[t];
    uint64_t keep = 0;
    while (m) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      const int g = t * 64 + b;
      if (g >= block_until) {
        keep |= 1ull << b;
        block_until = g + 4;
      }
    }
    out[t] = keep;
  }
}

template <typename R>
__device__ __forceinline__ double ld(const void* p, int64_t k) {
  return (double)reinterpret_cast<const R*>(p)[k];
}

// Response state of one lane at gap g for one chain.
struct Resp {
  double un, dn, us, ds;
  bool cum_i, cum_iv;
};

// sum over exposures r <= g of rho^(g-r) (and derivative), literal abd.py:258-274 restricted to set bits.
// tmax: number of 64-gap words to scan (wave-uniform); g: this lane's gap.
__device__ __forceinline__ Resp responses(int g, int tmax, const uint64_t I[ABD_MAXT], const uint64_t V[ABD_MAXT],
                                          const double2_t* tab_n, const double2_t* tab_s) {
  Resp r;
  r.un = r.dn = r.us = r.ds = 0.0;
  bool ci = false, civ = false;
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    if (t < tmax) {
      const int rel = g - t * 64;  // bits <= rel of this word are exposures at or before g
      const uint64_t le = rel >= 63 ? ~0ull : (rel < 0 ? 0ull : ((2ull << rel) - 1ull));
      ci |= (I[t] & le) != 0;
      civ |= ((I[t] | V[t]) & le) != 0;
      uint64_t m = I[t];
      while (m) {  // wave-uniform loop
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        int idx = g - (t * 64 + b) + 1;
        idx = idx < 0 ? 0 : idx;
        const double2_t pn = tab_n[idx];
        const double2_t ps = tab_s[idx];
        r.un += pn.x;
        r.dn += pn.y;
        r.us += ps.x;
        r.ds += ps.y;
      }
      m = V[t];
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        int idx = g - (t * 64 + b) + 1;
        idx = idx < 0 ? 0 : idx;
        const double2_t ps = tab_s[idx];
        r.us += ps.x;
        r.ds += ps.y;
      }
    }
  }
  r.cum_i = ci;
  r.cum_iv = civ;
  return r;
}

// One observation of one antigen: logistic curve + Normal log-term + its raw gradient sums.
//   a: inflection titer at this lane's (gap, ind); x: log_dilution; y: od     abd.py:459-469, 556-557
template <bool GRAD>
__device__ __forceinline__ void obs_term(double a, double x, double y, double b, double d, double isig,
                                         double& r2, double& sh, double& shx, double& sws, double& h_out) {
  const double amx = a - x;
  const double u = b * amx;               // -b (x - a)
  const double e = exp(u);
  const double s = 1.0 / (1.0 + e);       // logistic / d
  const double r = (y - d * s) * isig;
  r2 = fma(r, r, r2);
  if (GRAD) {
    const double w = r * isig;            // d ll / d m
    const double oms = e * s;             // 1 - s
    const double h = (w * d) * (s * oms);
    sh += h;
    shx = fma(h, amx, shx);
    sws = fma(w, s, sws);
    h_out = h;
  }
}

template <typename R, int CPW, bool DENSE, bool GRAD>
__global__ __launch_bounds__(ABD_BLOCK) void abd_eval_kernel(const EvalArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // LDS: [CPW][2][G+1] power tables + [G+1] "ones" table (non-waners: rho_j = 1) + block reduction
  double2_t* tabs = reinterpret_cast<double2_t*>(smem_raw);
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
    fill_pow_table(tabs + (c * 2 + 0) * tstride, a.ch[cbase + c].rho_n, G, tid, ABD_BLOCK);
    fill_pow_table(tabs + (c * 2 + 1) * tstride, a.ch[cbase + c].rho_s, G, tid, ABD_BLOCK);
  }
  for (int e = tid; e <= G; e += ABD_BLOCK) {
    double2_t v;
    v.x = e == 0 ? 0.0 : 1.0;
    v.y = 0.0;
    tab_ones[e] = v;
  }
  __syncthreads();

  double acc[CPW][ABD_NACC];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int k = 0; k < ABD_NACC; ++k) acc[c][k] = 0.0;
  int n1[CPW], m1[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) n1[c] = m1[c] = 0;

  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    // ---- integer pre-pass: indicator rows -> masks (ballot), constrain on the scalar unit ----
    uint64_t V[ABD_MAXT], P[ABD_MAXT], I[CPW][ABD_MAXT];
    const int64_t row = (int64_t)j * G;
    {
      uint64_t Rw[CPW][ABD_MAXT];
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) {
        V[t] = 0;
        P[t] = 0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) Rw[c][t] = 0;
        if (t < nt) {
          const int g = t * 64 + lane;
          const bool in = g < G;
          const int8_t vb = in ? a.vacs[row + g] : 0;
          const int8_t pb = (in && a.pcr) ? a.pcr[row + g] : 0;
          V[t] = __ballot(vb != 0);
          P[t] = __ballot(pb != 0);
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const int8_t rb = in ? a.ch[cbase + c].iraw[row + g] : 0;
            Rw[c][t] = __ballot(rb != 0);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        constrain_masks(Rw[c], P, a, I[c]);
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t) n1[c] += __builtin_popcountll(Rw[c][t]);  // Bernoulli(i_raw|p) is on the RAW matrix
      }
    }
    int wj[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      wj[c] = a.ch[cbase + c].waner[j] != 0;
      m1[c] += wj[c];
    }

    if (DENSE) {
      // ---- dense panel: one S and one N reading per cell, k = j*G + g ----
      for (int t = 0; t < nt; ++t) {
        const int g = t * 64 + lane;
        const bool in = g < G;
        const int64_t k = row + (in ? g : 0);
        const double yn = ld<R>(a.y_n, k), xn = ld<R>(a.x_n, k);
        const double ys = ld<R>(a.y_s, k), xs = ld<R>(a.x_s, k);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const ChainPar& p = a.ch[cbase + c];
          const double2_t* tn = tabs + (c * 2 + 0) * tstride;
          const double2_t* ts = wj[c] ? tabs + (c * 2 + 1) * tstride : tab_ones;
          const Resp rs = responses(g, t + 1, I[c], V, tn, ts);
          if (in) {
            // mu_n = perm + temp + init   abd.py:341 ; mu_s = perm + tinf + tvac + init   abd.py:389-391
            const double an = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
            const double as = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
            double h = 0.0;
            obs_term<GRAD>(an, xn, yn, p.b_n, p.d_n, p.isig_n, acc[c][A_N_R2], acc[c][A_N_H], acc[c][A_N_HX],
                           acc[c][A_N_WS], h);
            if (GRAD) {
              acc[c][A_N_HC] += rs.cum_i ? h : 0.0;
              acc[c][A_N_HU] = fma(h, rs.un, acc[c][A_N_HU]);
              acc[c][A_N_HD] = fma(h, rs.dn, acc[c][A_N_HD]);
            }
            obs_term<GRAD>(as, xs, ys, p.b_s, p.d_s, p.isig_s, acc[c][A_S_R2], acc[c][A_S_H], acc[c][A_S_HX],
                           acc[c][A_S_WS], h);
            if (GRAD) {
              acc[c][A_S_HC] += rs.cum_iv ? h : 0.0;
              acc[c][A_S_HD] = fma(h, rs.ds, acc[c][A_S_HD]);
            }
          }
        }
      }
    } else {
      // ---- sparse observation lists, CSR by individual (the real cohorts; abd.py:343, 393) ----
#pragma unroll
      for (int ag = 0; ag < 2; ++ag) {
        const int32_t* ptr = ag == 0 ? a.ptr_n : a.ptr_s;
        const uint8_t* gi = ag == 0 ? a.g_n : a.g_s;
        const void* yy = ag == 0 ? a.y_n : a.y_s;
        const void* xx = ag == 0 ? a.x_n : a.x_s;
        const int k0 = ptr[j], k1 = ptr[j + 1];
        for (int kb = k0; kb < k1; kb += 64) {
          const int k = kb + lane;
          const bool in = k < k1;
          const int kk = in ? k : k0;
          const int g = gi[kk];
          const double y = ld<R>(yy, kk), x = ld<R>(xx, kk);
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const ChainPar& p = a.ch[cbase + c];
            const double2_t* tn = tabs + (c * 2 + 0) * tstride;
            const double2_t* ts = wj[c] ? tabs + (c * 2 + 1) * tstride : tab_ones;
            const Resp rs = responses(g, nt, I[c], V, tn, ts);
            if (in) {
              double h = 0.0;
              if (ag == 0) {
                const double an = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
                obs_term<GRAD>(an, x, y, p.b_n, p.d_n, p.isig_n, acc[c][A_N_R2], acc[c][A_N_H], acc[c][A_N_HX],
                               acc[c][A_N_WS], h);
                if (GRAD) {
                  acc[c][A_N_HC] += rs.cum_i ? h : 0.0;
                  acc[c][A_N_HU] = fma(h, rs.un, acc[c][A_N_HU]);
                  acc[c][A_N_HD] = fma(h, rs.dn, acc[c][A_N_HD]);
                }
              } else {
                const double as = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
                obs_term<GRAD>(as, x, y, p.b_s, p.d_s, p.isig_s, acc[c][A_S_R2], acc[c][A_S_H], acc[c][A_S_HX],
                               acc[c][A_S_WS], h);
                if (GRAD) {
                  acc[c][A_S_HC] += rs.cum_iv ? h : 0.0;
                  acc[c][A_S_HD] = fma(h, rs.ds, acc[c][A_S_HD]);
                }
              }
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
      red[(wave * CPW + c) * ABD_NOUT + ABD_NACC] = (double)n1[c];
      red[(wave * CPW + c) * ABD_NOUT + ABD_NACC + 1] = (double)m1[c];
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

// Fixed-order sum of the per-block partials: one block per chain.  out[chain][ABD_NOUT] may live in
// mapped host memory (the 16 doubles per chain are the only thing that crosses PCIe per evaluation).
__global__ __launch_bounds__(256) void abd_finalize_kernel(const double* __restrict__ partials, int n_blocks,
                                                           double* __restrict__ out) {
  __shared__ double sm[16][ABD_NOUT];
  const int chain = blockIdx.x;
  const int k = threadIdx.x % ABD_NOUT;
  const int part = threadIdx.x / ABD_NOUT;  // 0..15
  const double* p = partials + (int64_t)chain * n_blocks * ABD_NOUT;
  double v = 0.0;
  for (int b = part; b < n_blocks; b += 16) v += p[(int64_t)b * ABD_NOUT + k];
  sm[part][k] = v;
  __syncthreads();
  if (threadIdx.x < ABD_NOUT) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sm[q][threadIdx.x];
    out[chain * ABD_NOUT + threadIdx.x] = s;
  }
}

// (G, N) gap-major int8 (PyMC's i_raw) -> (N, G) individual-major, 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void abd_transpose_i8_kernel(const int8_t* __restrict__ src, int8_t* __restrict__ dst,
                                                               int G, int N) {
  __shared__ int8_t tile[64][65];
  const int g0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int g = g0 + r, j = j0 + tx;
    tile[r][tx] = (g < G && j < N) ? src[(int64_t)g * N + j] : 0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int j = j0 + r, g = g0 + tx;
    if (g < G && j < N) dst[(int64_t)j * G + g] = tile[tx][r];
  }
}

__global__ void abd_flip_kernel(int8_t* iraw_ng, int8_t* waner, int G, int N, int64_t flat) {
  const int64_t gn = (int64_t)G * N;
  if (flat < gn) {
    const int64_t g = flat / N, j = flat % N;
    iraw_ng[j * G + g] ^= 1;
  } else {
    waner[flat - gn] ^= 1;
  }
}

// Deterministics "i", "ab_n_mu", "ab_s_mu" for one chain, written (G, N) gap-major as PyMC records them.
__global__ __launch_bounds__(ABD_BLOCK) void abd_deterministics_kernel(const EvalArgs a, int8_t* __restrict__ out_i,
                                                                       double* __restrict__ out_mun,
                                                                       double* __restrict__ out_mus) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double2_t* tabs = reinterpret_cast<double2_t*>(smem_raw);
  const int G = a.G, N = a.N, nt = a.nt, tstride = G + 1;
  double2_t* tab_ones = tabs + 2 * tstride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ChainPar& p = a.ch[0];
  fill_pow_table(tabs, p.rho_n, G, tid, ABD_BLOCK);
  fill_pow_table(tabs + tstride, p.rho_s, G, tid, ABD_BLOCK);
  for (int e = tid; e <= G; e += ABD_BLOCK) {
    double2_t v;
    v.x = e == 0 ? 0.0 : 1.0;
    v.y = 0.0;
    tab_ones[e] = v;
  }
  __syncthreads();
  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    uint64_t V[ABD_MAXT], P[ABD_MAXT], Rw[ABD_MAXT], I[ABD_MAXT];
    const int64_t row = (int64_t)j * G;
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) {
      V[t] = P[t] = Rw[t] = 0;
      if (t < nt) {
        const int g = t * 64 + lane;
        const bool in = g < G;
        V[t] = __ballot((in ? a.vacs[row + g] : 0) != 0);
        P[t] = __ballot(((in && a.pcr) ? a.pcr[row + g] : 0) != 0);
        Rw[t] = __ballot((in ? p.iraw[row + g] : 0) != 0);
      }
    }
    constrain_masks(Rw, P, a, I);
    const bool wj = p.waner[j] != 0;
    const double2_t* ts = wj ? tabs + tstride : tab_ones;
    for (int t = 0; t < nt; ++t) {
      const int g = t * 64 + lane;
      if (g < G) {
        const Resp rs = responses(g, t + 1, I, V, tabs, ts);
        const int64_t o = (int64_t)g * N + j;
        if (out_i) out_i[o] = (int8_t)((I[t] >> lane) & 1ull);
        if (out_mun) out_mun[o] = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
        if (out_mus) out_mus[o] = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
      }
    }
  }
}
