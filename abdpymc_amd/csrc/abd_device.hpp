// abd_device.hpp -- device helpers shared by every kernel family: wave reductions, power tables, the integer
// pre-pass (abd.py:640-667) on packed words, the logistic observation terms (abd.py:556-557, 459-469), the
// fixed-order sum of per-block partials.
#pragma once

#include "abd_types.hpp"

// force a (wave-uniform) value into vector registers
__device__ __forceinline__ double to_vgpr(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

// A workgroup counts in at an agent-scope counter on behalf of the rows it has stored (the stores of all its waves lie behind
// a workgroup barrier the caller has passed; the rows themselves are write-through (sc1) stores, re-read with sc1 loads).
//   ABD_HANDOFF_FORMAL = 0 (the product): the form the MI355X guide lists as valid and measured on gfx950 -- the storing wave
//     drains its stores (s_waitcnt vmcnt(0)), then a relaxed returning agent-scope add; the workgroup whose add came last
//     re-reads the rows with sc1 loads behind a barrier.  It rests on gfx9's store acknowledgements (a write-through store
//     is acknowledged once it is visible at agent scope), not on the HIP memory model: gfx950 builds only (checked below).
//   ABD_HANDOFF_FORMAL = 1: the counting add is acq_rel at agent scope (release/acquire as the memory model defines them, no
//     inline assembly); = 2: release with the add, acquire only in the workgroup that turns out to be last.
//     Measured (round 4, profiles/README.md: g_handoff_ab.txt): both cost far more than the 3 % that would have been accepted --
//     every workgroup's release is a buffer_wbl2 sc1 of its XCD's L2 -- config 3, evaluations/s seen by NUTS while all four
//     chains are at work: 141 k (0) / 48 k (1) / 42 k (2); default cohort 285 k / 221 k / 246 k.  Hence 0.
#ifndef ABD_HANDOFF_FORMAL
#define ABD_HANDOFF_FORMAL 0
#endif
#if ABD_HANDOFF_FORMAL == 0 && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the drained-stores hand-off (ABD_HANDOFF_FORMAL=0) is validated on gfx950 only: build other targets with -DABD_HANDOFF_FORMAL=1"
#endif
__device__ __forceinline__ unsigned int handoff_count_in(unsigned int* cnt) {
#if ABD_HANDOFF_FORMAL == 1
  return __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#elif ABD_HANDOFF_FORMAL == 2
  return __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
  return __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// the workgroup whose add came last, before it reads the other workgroups' rows
__device__ __forceinline__ void handoff_acquire() {
#if ABD_HANDOFF_FORMAL == 2
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
// the storing wave, all lanes, between its stores and the lane that counts in for them
__device__ __forceinline__ void handoff_drain_stores() {
#if ABD_HANDOFF_FORMAL
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the lanes' stores happen before the counting lane's release
  __builtin_amdgcn_wave_barrier();
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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
template <int MT, typename ARGS>  // MT: 64-gap words per individual; ARGS: anything with n_chunks and chunk_mask
__device__ __forceinline__ void constrain_masks(const uint64_t (&raw)[MT], const uint64_t (&pcr)[MT],
                                                const ARGS& a, uint64_t (&out)[MT]) {
  uint64_t i0[MT];
  if (a.n_chunks <= 1) {
    // OneTimeChunk: where(i_raw + pcrpos > 0, 1, 0)   abd.py:643-647
#pragma unroll
    for (int t = 0; t < MT; ++t) i0[t] = raw[t] | pcr[t];
  } else {
#pragma unroll
    for (int t = 0; t < MT; ++t) i0[t] = 0;
    for (int c = 0; c < a.n_chunks; ++c) {
      // mask_multiple_infections on the chunk: keep the first 1   abd.py:818
      // incorporate_pcrpos: any PCR+ in the chunk replaces the whole chunk column   abd.py:771
      bool has_pcr = false;
#pragma unroll
      for (int t = 0; t < MT; ++t) has_pcr |= (pcr[t] & a.chunk_mask[c][t]) != 0;
      bool found = false;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
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
  for (int t = 0; t < MT; ++t) {
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
// sum_chain_coherent leaves the 16 sums in sm[0 .. 15] (every thread may read them after it returns).
template <int NT>
__device__ __forceinline__ void sum_chain_coherent(const double* p, int n_blocks, double* sm, int tid) {
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
}
template <int NT>
__device__ __forceinline__ void finalize_chain_coherent(const double* p, int n_blocks, double* out, double* sm, int tid, double tag) {
  sum_chain_coherent<NT>(p, n_blocks, sm, tid);
  if (tid == 0) {
#pragma unroll
    for (int q = 0; q < ABD_NOUT - 1; ++q) out[q] = sm[q];
    __threadfence_system();
    __hip_atomic_store(out + (ABD_NOUT - 1), tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the double whose high word is `hi` and whose low word is `zero` (a register that holds 0; 0.0 and 1.0 are such
// doubles).  Each stream of such doubles gets its own zero register (zero_vgpr), so that the register pair is
// {that register, hi} and the high word is computed in place -- with a shared literal 0 the compiler copies it
// into the low half of every new pair, one v_mov_b32 per double per gap.
__device__ __forceinline__ double hi_to_double(uint32_t hi, uint32_t zero) { return __hiloint2double((int)hi, (int)zero); }
__device__ __forceinline__ uint32_t zero_vgpr() {
  uint32_t z = 0;
  asm volatile("" : "+v"(z));
  return z;
}
// rho * x + y with a scalar rho as ONE three-address v_fma_f64 (left to itself the compiler picks the two-address
// v_fmac_f64 for the loop-carried recurrences and pays a v_mov_b64 to keep the old value)
__device__ __forceinline__ double fma_s(double rho_sgpr, double x, double y) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(rho_sgpr), "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ double fma_v(double rho_vgpr, double x, double y) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(rho_vgpr), "v"(x), "v"(y));
  return r;
}

// 1 + 2^(t1024 / 1024).  kf = rint(t1024), f = t1024 - kf exact; v_cvt_i32_f64 saturates, so any finite t1024 gives
// a finite result (the exponent is clamped to [-1022, 510]: 2^510 keeps the product of the two antigens' terms
// finite, below 2^-1022 the term is 1 anyway); NaN stays NaN.
__device__ __forceinline__ double one_plus_exp2_tab(double t1024, const double* tab /* LDS */) {
  const double kf = __builtin_rint(t1024);
  const double f = t1024 - kf;
  int k;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(kf));  // saturating; a C++ cast of an out-of-range double is undefined
  const double T = tab[k & (ABD_EXP2_TAB - 1)];
  const int e = min(max(k >> 10, -1022), 510);  // v_med3_i32
  const double Ts = __hiloint2double(__double2hiint(T) + (e << 20), __double2loint(T));  // T 2^e: v_lshl_add_u32
  double p = fma(0x1.c6b08d910ecbdp-35, f, 0x1.ebfbe033445b4p-23);  // tools/exp2_table.py 1024 3
  p = fma(p, f, 0x1.62e42fefa39efp-11);
  p = fma(p, f, 1.0);
  return fma(Ts, p, 1.0);
}
static_assert(ABD_EXP2_TAB == 1024, "one_plus_exp2_tab: k >> 10 and the polynomial assume 1024 entries");

struct Resp {
  double un, dn, us, ds;
  bool cum_i, cum_iv;
};

// sum over exposures r <= g of rho^(g-r) (and derivative), literal abd.py:258-274 restricted to set bits.
template <int MT>
__device__ __forceinline__ Resp responses(int g, int tmax, const uint64_t (&I)[MT], const uint64_t (&V)[MT],
                                          const double2_t* tab_n, const double2_t* tab_s) {
  Resp r;
  r.un = r.dn = r.us = r.ds = 0.0;
  bool ci = false, civ = false;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
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


__device__ __forceinline__ double readfirstlane_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ---- wave sum without LDS traffic: four DPP steps inside each row of 16 lanes, then the four row totals ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_uniform(double v) {  // the same value in every lane (wave-uniform)
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

