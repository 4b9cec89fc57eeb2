// abd_kernels.hpp -- gfx950 (CDNA4) device code of the abdpymc joint-logp hot path.
//
// What the reference computes per evaluation (abdpymc/abd.py) and how it is mapped here:
//
//   i = constrain(i_raw, pcrpos)          abd.py:640-667, 560-601, 732-818   -> 64-bit words, bit b of word w = gap 64 w + b
//   perm_response / temp responses         abd.py:242-306                     -> recurrence (abd.py:277-293) inside a gap
//                                                                               segment, closed form (abd.py:258-274) to enter it
//   mu[idx_gap, idx_ind] -> logistic -> N  abd.py:343, 393, 445-469, 556-557  -> per-lane fp64, register sums
//
// Indicator panels (i_raw per chain, pcrpos, vacs) are held bit-packed, one 64-bit word per individual
// per 64 gaps, laid out [word][individual].
//
// Dense panels (benchmark configs; exactly one S and one N reading per cell): the OD panels are held
// gap-major, (G, N) arrays of [od, log_dilution] pairs -- the reference's own (gap, ind) orientation.
// A LANE owns one individual and walks consecutive gaps, so a wave's load of one gap row is 64 contiguous
// pairs (1 KiB), there is no cross-lane traffic until the final reduction, and no lane is idle whatever
// G is.  The (64-individual group, gap) plane is cut into equal ranges, one per wave slot; the state a
// range starts from (titer responses, their rho-sensitivities, exposure flags) is rebuilt per lane from
// the packed words (abd_dense.hpp).  The 4 waves of a workgroup are 4 chains of the same range, so the
// panel rows they share are fetched from HBM once and served from L1/L2 to the others.
//
// Sparse observation lists (the real cohorts): one wave per individual, lanes over its observations
// (CSR by individual), same masks, responses by the closed form.
//
// Sums stay in registers, are reduced once per wave (halving butterfly), once per block through LDS,
// written as per-block partials and summed in a fixed order by a second tiny kernel: no float atomics,
// bit-reproducible for a given launch shape.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ABD_MAXT 4          // 64-gap words per individual (G <= 256)
#define ABD_NACC 13         // floating sums per chain (see enum below)
#define ABD_NOUT 16         // ABD_NACC + n1 + m1, padded
#define ABD_MAX_BATCH_K 16  // chains per launch
#define ABD_WAVES_PER_BLOCK 4
#define ABD_BLOCK (64 * ABD_WAVES_PER_BLOCK)

// raw sums accumulated on the device; per-chain constants are applied on the host (abd_capi.hip: assemble)
//   q = od - d s,  s = 1 / (1 + exp(b (a - x))),  h' = q s (1 - s)
//   d ll / d a = -b (d / sigma^2) h'
enum {
  A_N_Q2 = 0,   // sum q^2                        -> ll, d/d log sigma
  A_N_H,        // sum h'            -> init_n
  A_N_HC,       // sum h' [cumI>0]   -> perm_n
  A_N_HU,       // sum h' U_n        -> temp_n    U_n = sum_r rho_n^(g-r)
  A_N_HD,       // sum h' dU_n/drho  -> rho_n
  A_N_HX,       // sum h' (a - x)    -> b_n
  A_N_QS,       // sum q s           -> d_n
  A_S_Q2,
  A_S_H,
  A_S_HC,
  A_S_HD,
  A_S_HX,
  A_S_QS,
};

struct ChainPar {
  double perm_n, temp_n, rho_n, init_n;
  double perm_s, rho_s, init_s;
  double b_n, d_n;
  double b_s, d_s;
  const uint64_t* rw;   // [nt][N] packed i_raw of the chain
  const int8_t* waner;  // [N]
};

struct EvalArgs {
  // sparse observation lists (CSR by individual)
  const void* y_n;
  const void* x_n;
  const void* y_s;
  const void* x_s;
  const uint8_t* g_n;
  const uint8_t* g_s;
  const int32_t* ptr_n;
  const int32_t* ptr_s;
  const int32_t* j_n;  // individual of each observation (observation-lane kernel)
  const int32_t* j_s;
  int32_t K_n, K_s;          // list lengths
  int32_t ob_n, ob_s, ob_c;  // observation-lane kernel: workgroups over the N list, the S list, the individuals
  int32_t pad3_;
  // dense panels, gap-major [G][N] of {od, log_dilution}
  const void* yx_n;
  const void* yx_s;
  // packed indicator panels [nt][N]
  const uint64_t* vw;
  const uint64_t* pw;  // nullptr = ignore_pcrpos
#ifdef ABD_STAMPS
  unsigned long long* stamps;  // diagnostic build: [grid.x][16] s_memrealtime at phase boundaries
#endif
  const int32_t* range_tab;  // dense kernel: {first lane group, first gap, rows, 0} of every range of this launch shape
  const double* exp2_tab;  // dense kernel: 2^(j/1024), j = 0..1023, correctly rounded (copied to LDS per workgroup)
  double* partials;    // [n_chains][grid.x][ABD_NOUT]
  // dense kernel only: the fixed-order sum of the PREVIOUS launch's partials, done by the first
  // prev_n_chains workgroups of this launch (saves a kernel and a boundary per step when launches are
  // stream-ordered); prev_n_chains = 0 -> nothing to do
  const double* prev_partials;
  double* prev_out;
  int32_t prev_n_chains, prev_blocks;
  int32_t fin_rows;    // gap rows the finalizing workgroups are excused from
  int32_t xcd_remap;   // dense kernel: workgroup -> range mapping that keeps neighbouring ranges on one XCD
  double prev_tag;          // completion tag of the previous launch (see finalize_chain)
  // observation-lane kernel: the workgroup of a chain that finishes last sums that chain's partial rows itself
  // (abd_obs.hpp) -- one launch per evaluation instead of two.  fin_count: one zeroed counter per grid row.
  unsigned int* fin_count;
  double* fin_out;
  double fin_tag;
  int32_t G, N, nt, n_chunks;
  int32_t n_chains, n_lg;     // n_lg: 64-individual lane groups (dense kernel)
  uint64_t chunk_mask[3][ABD_MAXT];
  ChainPar ch[ABD_MAX_BATCH_K];
};

struct double2_t {
  double x, y;
};

template <typename R>
struct YX {
  R y, x;
};

// force a (wave-uniform) value into vector registers
__device__ __forceinline__ double to_vgpr(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum 16 per-lane values over the 64 lanes with a halving butterfly: after it, lane l with (l & 3) == 0
// holds the total of value index 8 b5 + 4 b4 + 2 b3 + b2 (b_k = bit k of l).  15 + 2 exchanges instead of 96.
template <int OFF, int HALF>
__device__ __forceinline__ void reduce16_step(double (&v)[16], int lane) {
  // compile-time indices only: a runtime-indexed register array becomes a 16-way select network
  const bool up = (lane & OFF) != 0;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const double send = up ? v[k] : v[k + HALF];
    const double keep = up ? v[k + HALF] : v[k];
    v[k] = keep + __shfl_xor(send, OFF, 64);
  }
}
// The same two first steps (lanes l <-> l ^ 32, then l <-> l ^ 16) with gfx950's v_permlane32_swap / v_permlane16_swap:
// swap(a, b) leaves {a.lo, b.lo} / {a.hi, b.hi} in the two registers (halves of 32 lanes; for the 16-lane form the odd
// rows of a and the even rows of b change places), so a + b IS the halving step -- 3 instructions per exchange of a
// double instead of 2 ds_bpermute + 4 selects + 1 add, and the sums are the same bits (a + b = b + a).
typedef unsigned int abd_u2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ double swap_add(double a, double b) {
  abd_u2 lo, hi;
  if (OFF == 32) {
    lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  } else {
    lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  }
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double wave_reduce16(double (&v)[16], int lane) {
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = swap_add<32>(v[k], v[k + 8]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = swap_add<16>(v[k], v[k + 4]);
  reduce16_step<8, 2>(v, lane);
  reduce16_step<4, 1>(v, lane);
  double r = v[0];
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}
__device__ __forceinline__ int reduce16_index(int lane) {
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

// Fill one power table of n_entries: tab[0] = {0,0} (index for "exposure is in the future"),
// tab[k+1] = {rho^k, k rho^(k-1)}.
__device__ __forceinline__ void fill_pow_table(double2_t* tab, double rho, int n_entries, int tid, int nthreads) {
  for (int e = tid; e < n_entries; e += nthreads) {
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

// Same table filled by ONE wave (dense kernel: each wave owns one chain).  Entry e >= 2 needs rho^(e-2):
// lane l holds z_l = rho^((l + 62) mod 64) from one 6-step binary powering, and entry e = 64 b + l is
// z_l times rho^(64 b) (lanes >= 2) or rho^(64 (b-1)) (lanes 0, 1) -- 3 multiplies per entry.
__device__ __forceinline__ void fill_pow_table_wave(double2_t* tab, double rho, int n_entries, int lane) {
  double base = rho, z = 1.0;
  int n = (lane + 62) & 63;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    if (n & 1) z *= base;
    base *= base;
    n >>= 1;
  }
  // base = rho^64
  double m_cur = 1.0, m_prev = 0.0;  // rho^(64 b), rho^(64 (b - 1))
  for (int b0 = 0; b0 < n_entries; b0 += 64) {
    const int e = b0 + lane;
    double2_t v;
    const double pkm1 = z * (lane < 2 ? m_prev : m_cur);  // rho^(e - 2)
    v.x = pkm1 * rho;
    v.y = (double)(e - 1) * pkm1;
    if (e == 0) {
      v.x = 0.0;  // index for "exposure is in the future"
      v.y = 0.0;
    } else if (e == 1) {
      v.x = 1.0;  // rho^0 = 1 also for rho = 0 (abd.py:258: rho**design with design = 0)
      v.y = 0.0;
    }
    if (e < n_entries) tab[e] = v;
    m_prev = m_cur;
    m_cur *= base;
  }
}

__device__ __forceinline__ void fill_ones_table_wave(double2_t* tab, int n_entries, int lane) {
  for (int e = lane; e < n_entries; e += 64) {
    double2_t v;
    v.x = e == 0 ? 0.0 : 1.0;
    v.y = 0.0;
    tab[e] = v;
  }
}

__device__ __forceinline__ void fill_ones_table(double2_t* tab, int n_entries, int tid, int nthreads) {
  for (int e = tid; e < n_entries; e += nthreads) {
    double2_t v;
    v.x = e == 0 ? 0.0 : 1.0;  // non-waners: rho_j = 1 (abd.py:374), d rho_j / d rho_s = 0
    v.y = 0.0;
    tab[e] = v;
  }
}

// Integer pre-pass for one individual and one chain (abd.py:640-667) on packed words.  Works equally on
// wave-uniform values (sparse kernel: scalar unit) and on per-lane values (dense kernel).
//   raw/pcr : words of i_raw / pcrpos, bit b of word t <-> gap 64 t + b
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

// 2^t for t <= 1021 (callers clamp): k = rint(t), f = t - k is exact, degree-10 near-minimax polynomial for
// 2^f on |f| <= 1/2 (tools/exp_poly.py 10 exp2: max relative error 3.1e-16), scale by 2^k.  14 VALU.
__device__ __forceinline__ double exp2_reduced(double t) {
  const double k = __builtin_rint(t);
  const double f = t - k;
  double p = 7.072585949269223e-09;
  p = fma(p, f, 1.0208690299958306e-07);
  p = fma(p, f, 1.321544258792169e-06);
  p = fma(p, f, 1.5252657260200837e-05);
  p = fma(p, f, 0.0001540353044173605);
  p = fma(p, f, 0.0013333558230164974);
  p = fma(p, f, 0.009618129107606888);
  p = fma(p, f, 0.05550410866444772);
  p = fma(p, f, 0.24022650695910097);
  p = fma(p, f, 0.69314718055995);
  p = fma(p, f, 1.0);
  return ldexp(p, (int)k);
}

// 1/d for d in [1, 2^1023): v_rcp_f64 seed (measured max rel. error 4.6e-8) + one Newton step
// (measured 2.2e-15; a second step gives 1.1e-16 for two more fma: tools/micro/rcp_accuracy.hip).
__device__ __forceinline__ double rcp_newton(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// One observation of one antigen: logistic curve (abd.py:556-557), residual of the Normal log-term
// (abd.py:459-469) and its raw gradient sums.  a: inflection titer at this (gap, ind); x: log_dilution; y: od.
template <bool GRAD, bool GUARD = true>
__device__ __forceinline__ void obs_term(double a, double x, double y, double b, double d, double guard, double& q2,
                                         double& sh, double& shx, double& sqs, double& h_out) {
  const double amx = a - x;
  // e = exp(-b (x - a)) = 2^t, t = (b log2 e)(a - x); clamped so that 1 + e stays finite (the curve is
  // ~1e-308 of d there anyway).  b log2 e is wave-uniform and hoisted out of the gap loop.
  const double t = fmin((b * 1.4426950408889634074) * amx, 1021.0);
  const double e = exp2_reduced(t);
  const double s = rcp_newton(1.0 + e);    // logistic / d
  double q = fma(-d, s, y);
  if (GUARD) q *= guard;  // guard = 0 on padding lanes, else 1
  q2 = fma(q, q, q2);
  if (GRAD) {
    const double u = q * s;
    sqs += u;
    const double h = fma(-u, s, u);  // q s (1 - s)
    sh += h;
    shx = fma(h, amx, shx);
    h_out = h;
  }
}

// Both antigens of one cell at once (dense panels): the two reciprocals 1/(1+e_n), 1/(1+e_s) come from ONE
// v_rcp_f64 (quarter rate) of the product -- 1/A = B/(AB), 1/B = A/(AB) -- which saves three issue slots per
// cell.  The exponents are clamped at 2^510 so that the product stays finite (the curve is ~1e-154 of d there).
template <bool GRAD>
__device__ __forceinline__ void obs_pair(double an, double xn, double yn, double b_n, double d_n, double as, double xs,
                                         double ys, double b_s, double d_s, double (&acc)[16], double& h_n, double& h_s) {
  const double amx_n = an - xn, amx_s = as - xs;
  const double e_n = exp2_reduced(fmin((b_n * 1.4426950408889634074) * amx_n, 510.0));
  const double e_s = exp2_reduced(fmin((b_s * 1.4426950408889634074) * amx_s, 510.0));
  const double A = 1.0 + e_n, B = 1.0 + e_s;
  const double r = rcp_newton(A * B);
  const double s_n = r * B, s_s = r * A;  // logistic / d
  const double q_n = fma(-d_n, s_n, yn), q_s = fma(-d_s, s_s, ys);
  acc[A_N_Q2] = fma(q_n, q_n, acc[A_N_Q2]);
  acc[A_S_Q2] = fma(q_s, q_s, acc[A_S_Q2]);
  if (GRAD) {
    const double u_n = q_n * s_n, u_s = q_s * s_s;
    acc[A_N_QS] += u_n;
    acc[A_S_QS] += u_s;
    h_n = fma(-u_n, s_n, u_n);  // q s (1 - s)
    h_s = fma(-u_s, s_s, u_s);
    acc[A_N_H] += h_n;
    acc[A_S_H] += h_s;
    acc[A_N_HX] = fma(h_n, amx_n, acc[A_N_HX]);
    acc[A_S_HX] = fma(h_s, amx_s, acc[A_S_HX]);
  }
}

// ================================================================================================
// Dense-panel kernel: lane = individual, wave = (64 individuals, chain, gap segment)
// ================================================================================================

// Fixed-order sum of one chain's per-block partials by one workgroup of NT threads: 64 interleaved partial
// sums (block b goes to partial b mod 64; 16 independent loads in flight per thread), then a 64-way sum per
// value.  The order depends only on n_blocks -- not on NT -- so the standalone kernel (1024 threads) and the
// fused form inside the next dense launch (256 threads, abd_dense.hpp) give identical bits.
// sm: 64 x ABD_NOUT doubles of LDS scratch.  out may live in mapped host memory (the 16 doubles per chain
// are the only thing that crosses PCIe per evaluation).
#define ABD_FIN_PARTS 64
template <int NT>
__device__ __forceinline__ void finalize_chain(const double* __restrict__ p, int n_blocks, double* __restrict__ out,
                                               double* sm, int tid, double tag) {
  const int k = tid % ABD_NOUT;
  for (int part = tid / ABD_NOUT; part < ABD_FIN_PARTS; part += NT / ABD_NOUT) {
    double v = 0.0;
    for (int b0 = part; b0 < n_blocks; b0 += 16 * ABD_FIN_PARTS) {
      double q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = b0 + u * ABD_FIN_PARTS;
        q[u] = b < n_blocks ? p[(int64_t)b * ABD_NOUT + k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) v += q[u];
    }
    sm[part * ABD_NOUT + k] = v;
  }
  __syncthreads();
  double t = 0.0;
  if (tid < ABD_NOUT) {
#pragma unroll
    for (int q = 0; q < ABD_FIN_PARTS; ++q) t += sm[q * ABD_NOUT + tid];
  }
  __syncthreads();
  if (tid < ABD_NOUT) sm[tid] = t;
  __syncthreads();
  if (tid == 0) {
    // one lane writes the row, then -- behind a system-scope fence -- the launch's tag into the spare 16th
    // double: a host that polls the tag in mapped memory sees a complete row without a stream synchronise
#pragma unroll
    for (int q = 0; q < ABD_NOUT - 1; ++q) out[q] = sm[q];
    __threadfence_system();
    __hip_atomic_store(out + (ABD_NOUT - 1), tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// finalize_chain with device-coherent loads of the partial rows (they were written write-through by other workgroups of the
// SAME kernel, possibly on other XCDs: a plain load could hit a stale line of this XCD's L2).  Same order of additions, same bits.
template <int NT>
__device__ __forceinline__ void finalize_chain_coherent(const double* p, int n_blocks, double* out, double* sm, int tid, double tag) {
  const int k = tid % ABD_NOUT;
  for (int part = tid / ABD_NOUT; part < ABD_FIN_PARTS; part += NT / ABD_NOUT) {
    double v = 0.0;
    for (int b0 = part; b0 < n_blocks; b0 += 16 * ABD_FIN_PARTS) {
      double q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = b0 + u * ABD_FIN_PARTS;
        q[u] = b < n_blocks ? __hip_atomic_load(p + (int64_t)b * ABD_NOUT + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) v += q[u];
    }
    sm[part * ABD_NOUT + k] = v;
  }
  __syncthreads();
  double t = 0.0;
  if (tid < ABD_NOUT) {
#pragma unroll
    for (int q = 0; q < ABD_FIN_PARTS; ++q) t += sm[q * ABD_NOUT + tid];
  }
  __syncthreads();
  if (tid < ABD_NOUT) sm[tid] = t;
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int q = 0; q < ABD_NOUT - 1; ++q) out[q] = sm[q];
    __threadfence_system();
    __hip_atomic_store(out + (ABD_NOUT - 1), tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#include "abd_dense.hpp"
#include "abd_resident.hpp"

// ================================================================================================
// Sparse-list kernel (the real cohorts: several dilutions per serum sample, most cells empty)
// ================================================================================================

struct Resp {
  double un, dn, us, ds;
  bool cum_i, cum_iv;
};

// sum over exposures r <= g of rho^(g-r) (and derivative), literal abd.py:258-274 restricted to set bits.
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

// wave-uniform load of one packed word
__device__ __forceinline__ uint64_t uniform_word(const uint64_t* p, int64_t idx) {
  const uint64_t v = p[idx];
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

template <typename R, int CPW, bool GRAD>
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
  int n1[CPW], m1[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) n1[c] = m1[c] = 0;

  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    uint64_t V[ABD_MAXT], P[ABD_MAXT], I[CPW][ABD_MAXT];
    {
      uint64_t Rw[CPW][ABD_MAXT];
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) {
        V[t] = 0;
        P[t] = 0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) Rw[c][t] = 0;
        if (t < nt) {
          V[t] = uniform_word(a.vw, (int64_t)t * N + j);
          if (a.pw) P[t] = uniform_word(a.pw, (int64_t)t * N + j);
#pragma unroll
          for (int c = 0; c < CPW; ++c) Rw[c][t] = uniform_word(a.ch[cbase + c].rw, (int64_t)t * N + j);
        }
      }
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        constrain_masks(Rw[c], P, a, I[c]);
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t) n1[c] += __builtin_popcountll(Rw[c][t]);
      }
    }
    int wj[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      wj[c] = __builtin_amdgcn_readfirstlane((int)a.ch[cbase + c].waner[j]) != 0;
      m1[c] += wj[c];
    }

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
        const double guard = in ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const ChainPar& p = a.ch[cbase + c];
          const double2_t* tn = tabs + (c * 2 + 0) * tstride;
          const double2_t* ts = wj[c] ? tabs + (c * 2 + 1) * tstride : tab_ones;
          const Resp rs = responses(g, nt, I[c], V, tn, ts);
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

#include "abd_obs.hpp"
#include "abd_gibbs.hpp"
#include "abd_gibbs2.hpp"

// ================================================================================================
// Small kernels
// ================================================================================================

#define ABD_FIN_THREADS 1024
__global__ __launch_bounds__(ABD_FIN_THREADS) void abd_finalize_kernel(const double* __restrict__ partials, int n_blocks,
                                                                       double* __restrict__ out, double tag) {
  __shared__ double sm[ABD_FIN_PARTS * ABD_NOUT];
  finalize_chain<ABD_FIN_THREADS>(partials + (int64_t)blockIdx.x * n_blocks * ABD_NOUT, n_blocks,
                                  out + blockIdx.x * ABD_NOUT, sm, threadIdx.x, tag);
}

// One wave that stays on the device for `ticks` of the 100 MHz s_memrealtime counter and says when (abd_capi.hip:
// probe_stream_queues -- which of the context's HIP streams can have kernels on the device at the same time)
__global__ void abd_spin_kernel(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(16);
    t = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0) {
    out[0] = t0;
    out[1] = t;
  }
}

// The same copy by ONE workgroup, for small flushes, followed by a completion tag in mapped host memory (every thread
// fences its own stores at system scope before the barrier, thread 0 then releases the tag): abd_wait polls the tag
// instead of synchronising the stream.
__global__ __launch_bounds__(1024) void abd_copy_tag_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n,
                                                            double* done, double tag) {
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(done, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// device result ring -> mapped host memory, for stream-ordered launches (one flush per abd_wait)
__global__ void abd_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// (G, N) gap-major int8 (PyMC's i_raw, or vacs.T / pcrpos.T) -> packed words [nt][N]
__global__ __launch_bounds__(256) void abd_pack_bits_kernel(const int8_t* __restrict__ src, uint64_t* __restrict__ dst,
                                                            int G, int N, int nt) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (j >= N || t >= nt) return;
  uint64_t w = 0;
  const int g_end = min(64, G - t * 64);
  for (int b = 0; b < g_end; ++b) w |= (uint64_t)(src[(int64_t)(t * 64 + b) * N + j] != 0) << b;  // coalesced over j
  dst[(int64_t)t * N + j] = w;
}

__global__ void abd_flip_kernel(uint64_t* rw, int8_t* waner, int G, int N, int64_t flat) {
  const int64_t gn = (int64_t)G * N;
  if (flat < gn) {
    const int64_t g = flat / N, j = flat % N;
    rw[(g >> 6) * N + j] ^= 1ull << (g & 63);
  } else {
    waner[flat - gn] ^= 1;
  }
}

// Deterministics "i", "ab_n_mu", "ab_s_mu" for one chain, written (G, N) gap-major as PyMC records them;
// with `sums` ([3][G*N]: i, ab_n_mu, ab_s_mu) they are also added to running sums (posterior means on device).
__global__ __launch_bounds__(ABD_BLOCK) void abd_deterministics_kernel(const EvalArgs a, int8_t* __restrict__ out_i,
                                                                       double* __restrict__ out_mun,
                                                                       double* __restrict__ out_mus,
                                                                       double* __restrict__ sums) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  const int G = a.G, N = a.N, nt = a.nt, tstride = G + 1;
  double2_t* tab_ones = tabs + 2 * tstride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ChainPar& p = a.ch[0];
  fill_pow_table(tabs, p.rho_n, tstride, tid, ABD_BLOCK);
  fill_pow_table(tabs + tstride, p.rho_s, tstride, tid, ABD_BLOCK);
  fill_ones_table(tab_ones, tstride, tid, ABD_BLOCK);
  __syncthreads();
  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    uint64_t V[ABD_MAXT], P[ABD_MAXT], Rw[ABD_MAXT], I[ABD_MAXT];
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) {
      V[t] = P[t] = Rw[t] = 0;
      if (t < nt) {
        V[t] = uniform_word(a.vw, (int64_t)t * N + j);
        if (a.pw) P[t] = uniform_word(a.pw, (int64_t)t * N + j);
        Rw[t] = uniform_word(p.rw, (int64_t)t * N + j);
      }
    }
    constrain_masks(Rw, P, a, I);
    const bool wj = __builtin_amdgcn_readfirstlane((int)p.waner[j]) != 0;
    const double2_t* ts = wj ? tabs + tstride : tab_ones;
    for (int t = 0; t < nt; ++t) {
      const int g = t * 64 + lane;
      if (g < G) {
        const Resp rs = responses(g, t + 1, I, V, tabs, ts);
        const int64_t o = (int64_t)g * N + j;
        const int bit = (int)((I[t] >> lane) & 1ull);
        const double mun = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
        const double mus = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
        if (out_i) out_i[o] = (int8_t)bit;
        if (out_mun) out_mun[o] = mun;
        if (out_mus) out_mus[o] = mus;
        if (sums) {
          const int64_t cells = (int64_t)G * N;
          sums[o] += (double)bit;
          sums[cells + o] += mun;
          sums[2 * cells + o] += mus;
        }
      }
    }
  }
}
