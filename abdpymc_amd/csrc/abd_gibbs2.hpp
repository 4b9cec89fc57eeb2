// abd_gibbs2.hpp -- the binary Gibbs-Metropolis sweep for DENSE panels, speculative lane-per-proposal form
// (included by abd_gibbs.hip after abd_gibbs.hpp, whose Philox stream, GibbsArgs and helpers it shares).
//
// Same sweep as abd_gibbs_kernel (PyMC's BinaryGibbsMetropolis on [i_raw, ab_s_waner], abd.py:427, 373, 922; see
// abd_gibbs.hpp for why one individual's proposals can run on their own and in which random order), same random
// stream, same decisions -- a different mapping onto the wave:
//
//   abd_gibbs_kernel        one proposal at a time, LANES = the 64 gaps of a round; every proposal pays whole rounds,
//                           a wave-wide sum, and a pass over the packed words on the scalar unit
//   abd_gibbs_dense_kernel  LANES = PROPOSALS.  Almost every proposal is rejected (0.05 - 2 acceptances per individual
//                           and sweep), so the next proposals in the individual's random order are evaluated
//                           SPECULATIVELY against the current state, each by its own lane: the lane applies the
//                           constraints to its flipped column (abd.py:640-667; the three-gap pass restarts at the
//                           first changed gap), then walks gap by gap from the first gap whose constrained infection
//                           changes, carrying the two responses by the recurrence (abd.py:288) and adding up the new
//                           likelihood terms.  Every term is <= 0, so
//                               delta  <=  (prior delta - sum of the CURRENT terms from that gap on) + (new terms so far)
//                           and the lane stops as soon as that bound falls below log u -- typically 3-10 gaps after a
//                           new infection -- or at the last gap, where the bound IS delta.  Lanes that finish pick up
//                           the next proposals (refilled when >= 16 lanes are idle).  Results are committed strictly
//                           in the sweep's order: rejections advance a frontier; the first acceptance is applied
//                           (constraints and current terms recomputed by the whole wave, lanes = gaps) and everything
//                           evaluated beyond it is thrown away and redone against the new state.  The one
//                           ab_s_waner proposal changes every gap and is evaluated by the whole wave when the frontier
//                           reaches it.  The trajectory is therefore exactly the sequential one.
//
// Per wave in LDS: the individual's OD pairs of both antigens, the two responses at the current state per gap, the
// suffix sums of the current terms, log u per dim, the proposal list and one result byte per proposal.
#pragma once

#include "abd_gibbs.hpp"

#define ABD_G2_PENDING 0
#define ABD_G2_REJECT 1
#define ABD_G2_ACCEPT 2
#define ABD_G2_COMPLEX 3
#define ABD_G2_ITER_CAP (1 << 20)    // hard bound on scheduler iterations per individual (never reached: see the loop)

#define ABD_G2_MAX_WAVES 12  // waves of a workgroup (= of a CU: one workgroup per CU, three waves per SIMD, <= 168 registers)
__host__ __device__ inline size_t abd_g2_pad16(size_t b) { return (b + 15) / 16 * 16; }
// per-wave LDS bytes (9.6 KB at G = 200, fp64: LDS, not registers, decides how many waves a CU holds)
__host__ __device__ inline size_t abd_g2_wave_lds(int G, int rbytes) {
  size_t b = abd_g2_pad16((size_t)G * 2 * rbytes) * 2;  // {od, log_dilution} per gap, N then S
  b += abd_g2_pad16((size_t)(G + 1) * 8);                // suf[g] = -(sum of the current terms of gaps >= g); suf[G] = 0
  b += abd_g2_pad16((size_t)(G + 1) * 4);                // the acceptance draw's Philox word by dim
  b += abd_g2_pad16((size_t)(G + 1) * 2);                // proposal list: dim by position in the sweep
  b += abd_g2_pad16((size_t)(G + 1));                    // result by position
  return b;
}
// LDS of the tables every wave of the workgroup shares: [2][G+1] power tables + [G+1] ones + 2^(j/1024)
__host__ __device__ inline size_t abd_g2_shared_lds(int G) {
  return (size_t)3 * (G + 1) * sizeof(double2_t) + (size_t)ABD_EXP2_TAB * sizeof(double);
}
// waves per workgroup that fit the CU's 160 KB (0: not even one)
__host__ __device__ inline int abd_g2_waves(int G, int rbytes) {
  const size_t avail = (size_t)160 * 1024 - abd_g2_shared_lds(G);
  const size_t w = avail / abd_g2_wave_lds(G, rbytes);
  return (int)(w > ABD_G2_MAX_WAVES ? ABD_G2_MAX_WAVES : w);
}
__host__ __device__ inline size_t abd_g2_lds(int G, int rbytes, int n_waves) {
  return abd_g2_shared_lds(G) + (size_t)n_waves * abd_g2_wave_lds(G, rbytes);
}

// i0 of constrain_infections before the three-gap pass (abd.py:643-647 one chunk; abd.py:818 + 771 per chunk otherwise).
// Works on wave-uniform and on per-lane words alike.
__device__ __forceinline__ void constrain_i0(const uint64_t raw[ABD_MAXT], const uint64_t pcr[ABD_MAXT], const EvalArgs& a,
                                             uint64_t i0[ABD_MAXT]) {
  if (a.n_chunks <= 1) {
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) i0[t] = raw[t] | pcr[t];
  } else {
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) i0[t] = 0;
    for (int c = 0; c < a.n_chunks; ++c) {
      bool has_pcr = false;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) has_pcr |= (pcr[t] & a.chunk_mask[c][t]) != 0;
      bool found = false;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) {
        const uint64_t cm = a.chunk_mask[c][t];
        const uint64_t r = raw[t] & cm;
        const uint64_t first = found ? 0ull : (r & (0ull - r));
        found |= r != 0;
        i0[t] |= has_pcr ? (pcr[t] & cm) : first;
      }
    }
  }
}

// mask_three_gaps (abd.py:560-601) restarted at gap p0: the kept infections before p0 (`before` = I & bits below p0) are
// what they were -- the pass is causal -- and the greedy pass goes on from the last of them over the bits of i0 at >= p0.
__device__ __forceinline__ void three_gaps_from(const uint64_t i0[ABD_MAXT], const uint64_t before[ABD_MAXT], int p0,
                                                uint64_t out[ABD_MAXT]) {
  int block_until = 0;
#pragma unroll
  for (int t = ABD_MAXT - 1; t >= 0; --t)
    if (before[t] != 0 && block_until == 0) block_until = t * 64 + 63 - __builtin_clzll(before[t]) + 4;
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    const int rel = p0 - t * 64;  // bits >= rel of word t are at or after p0
    const uint64_t from = rel <= 0 ? ~0ull : (rel >= 64 ? 0ull : ~((1ull << rel) - 1ull));
    uint64_t m = i0[t] & from;
    uint64_t keep = before[t];
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

__device__ __forceinline__ int first_bit(const uint64_t w[ABD_MAXT]) {  // position of the lowest set bit, or 1 << 20
  int p = 1 << 20;
#pragma unroll
  for (int t = ABD_MAXT - 1; t >= 0; --t)
    if (w[t] != 0) p = t * 64 + __builtin_ctzll(w[t]);
  return p;
}

// word (g >> 6) of a per-lane or uniform row, g per lane; compile-time indices only
__device__ __forceinline__ uint64_t word_at(const uint64_t w[ABD_MAXT], int g) {
  const int t = g >> 6;
  uint64_t v = w[0];
#pragma unroll
  for (int q = 1; q < ABD_MAXT; ++q) v = t == q ? w[q] : v;
  return v;
}

// a wave-uniform row shifted down by g gaps (bit 0 of the result = gap g)
__device__ __forceinline__ void shift_row_down(const uint64_t w[ABD_MAXT], int g, uint64_t out[ABD_MAXT]) {
  const int q = g >> 6, sh = g & 63;
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < ABD_MAXT; ++k) {
      lo = t + q == k ? w[k] : lo;
      hi = t + q + 1 == k ? w[k] : hi;
    }
    out[t] = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
  }
}

// both antigens' likelihood term of one gap: -1/2 (q_n / sigma_n)^2 - 1/2 (q_s / sigma_s)^2 (terms that do not depend
// on the discrete state are left out: they cancel in every difference)
__device__ __forceinline__ double g2_term(double an, double xn, double yn, double c_n, double d_n, double nh_n, double as, double xs,
                                          double ys, double c_s, double d_s, double nh_s, const double* tab_e2) {
  const double A = one_plus_exp2_tab(c_n * (an - xn), tab_e2);
  const double B = one_plus_exp2_tab(c_s * (as - xs), tab_e2);
  const double r = rcp_newton(A * B);
  const double q_n = fma(-d_n, r * B, yn), q_s = fma(-d_s, r * A, ys);
  return fma(nh_s * q_s, q_s, (nh_n * q_n) * q_n);
}

// log((w + 1/2) / 2^32), the log of the acceptance uniform, to ~4e-15 absolute (the library log costs ~100 fp64
// instructions, five of them per lane and individual): v_log_f32 places the mantissa m in [1, 2) in bin j of the
// 2^(j/1024) table, m 2^(-j/1024) - 1 = r is tiny (the table read backwards is the reciprocal: 2^(-j/1024) =
// T[1024 - j] / 2) and log(1 + r) takes four terms.
__device__ __forceinline__ double log_uniform_u32(uint32_t w, const double* tab_e2 /* LDS */) {
  const double u = (double)w + 0.5;                                   // exact, in [0.5, 2^32)
  const double m = 2.0 * __builtin_amdgcn_frexp_mant(u);              // [1, 2)
  const int e2 = __builtin_amdgcn_frexp_exp(u) - 1;                   // u = m 2^e2
  int j = (int)__builtin_rintf(__builtin_amdgcn_logf((float)m) * 1024.0f);  // v_log_f32 = log2
  j = min(max(j, 0), 1024);
  const double inv_t = j == 0 ? 1.0 : 0.5 * tab_e2[1024 - j];         // 2^(-j/1024)
  const double r = fma(m, inv_t, -1.0);                               // |r| < 4e-4
  double pl = fma(r, -0.25, 1.0 / 3.0);
  pl = fma(pl, r, -0.5);
  pl = fma(pl, r, 1.0);
  pl *= r;                                                            // log(1 + r), r^5 / 5 < 2e-18 dropped
  return fma((double)((e2 - 32) * 1024 + j), 0.69314718055994530942 / 1024.0, pl);
}

struct G2Par {  // wave-uniform constants of the chain
  double perm_n, temp_n, rho_n, init_n, perm_s, rho_s, init_s, c_n, d_n, c_s, d_s, nh_n, nh_s;
};

// Whole-wave evaluation of one state, lanes = gaps of a round: the two responses at this lane's gap of every round
// (carry into the round x rho^(lane+1) + this round's exposures at or before the lane, power table) and the term there.
// The rounds start at gap g_off (0: the whole individual; otherwise I and V are the rows shifted down by g_off and
// cvn / cvs / ci / civ the state at gap g_off - 1: the rest of one lane's walk, taken over by the whole wave).
template <typename R>
__device__ __forceinline__ void g2_eval_rounds(const EvalArgs& a, const G2Par& p, int lane, const uint64_t I[ABD_MAXT],
                                               const uint64_t V[ABD_MAXT], const double2_t* tab_n, const double2_t* tab_s,
                                               double pwn, double pws, const YX<R>* dataN, const YX<R>* dataS,
                                               const double* tab_e2, double (&un_o)[ABD_MAXT], double (&us_o)[ABD_MAXT],
                                               double (&term_o)[ABD_MAXT], int g_off = 0, double cvn = 0.0, double cvs = 0.0,
                                               bool ci = false, bool civ = false) {
  // cvn / cvs: responses at the end of the previous round (wave-uniform)
  const uint64_t le = (2ull << lane) - 1ull;  // bits at or before this lane (lane 63: all ones)
  const int n_rounds = (a.G - g_off + 63) >> 6;
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    un_o[t] = us_o[t] = term_o[t] = 0.0;
    if (t < n_rounds) {
      double un = pwn * cvn, us = pws * cvs;
      uint64_t m = I[t];
      while (m) {  // wave-uniform loop over this word's infections
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const int idx = min(max(lane - b + 1, 0), a.G);  // 0 = "in the future"
        un += tab_n[idx].x;
        us += tab_s[idx].x;
      }
      m = V[t];
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        us += tab_s[min(max(lane - b + 1, 0), a.G)].x;
      }
      const bool cum_i = ci || (I[t] & le) != 0;
      const bool cum_iv = civ || ((I[t] | V[t]) & le) != 0;
      const int g = g_off + t * 64 + lane;
      const bool valid = g < a.G;
      const int gg = valid ? g : 0;
      const YX<R> on = dataN[gg], os = dataS[gg];
      const double an = p.init_n + (cum_i ? p.perm_n : 0.0) + p.temp_n * un;
      const double as = p.init_s + (cum_iv ? p.perm_s : 0.0) + us;
      const double term = g2_term(an, (double)on.x, (double)on.y, p.c_n, p.d_n, p.nh_n, as, (double)os.x, (double)os.y, p.c_s,
                                  p.d_s, p.nh_s, tab_e2);
      un_o[t] = un;
      us_o[t] = us;
      term_o[t] = valid ? term : 0.0;
      cvn = readlane_f64(un, 63);
      cvs = readlane_f64(us, 63);
      ci |= I[t] != 0;
      civ |= (I[t] | V[t]) != 0;
    }
  }
}

// compare-exchange step of the bitonic network between lanes l and l ^ J (same register)
template <int J>
__device__ __forceinline__ uint32_t bitonic_lane_step(uint32_t key, int lane, bool ascending) {
  const uint32_t other = (uint32_t)__shfl_xor((int)key, J, 64);
  const bool lower = (lane & J) == 0;
  return (lower == ascending) ? min(key, other) : max(key, other);
}
template <int K, int J>
__device__ __forceinline__ void bitonic_stage(uint32_t (&k)[4], int lane) {
  // element e = r * 64 + lane; direction of its sub-sequence: ascending iff (e & K) == 0
  if (J >= 64) {
    constexpr int dr = J / 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if ((r & dr) == 0) {
        const bool asc = K >= 256 ? true : ((r * 64) & K) == 0;
        const uint32_t lo = min(k[r], k[r | dr]), hi = max(k[r], k[r | dr]);
        k[r] = asc ? lo : hi;
        k[r | dr] = asc ? hi : lo;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool asc = K >= 256 ? true : (K >= 64 ? ((r * 64) & K) == 0 : (lane & K) == 0);
      k[r] = bitonic_lane_step<(J < 64 ? J : 1)>(k[r], lane, asc);
    }
  }
}
template <int K, int J>
struct BitonicJ {
  static __device__ __forceinline__ void run(uint32_t (&k)[4], int lane) {
    bitonic_stage<K, J>(k, lane);
    BitonicJ<K, J / 2>::run(k, lane);
  }
};
template <int K>
struct BitonicJ<K, 0> {
  static __device__ __forceinline__ void run(uint32_t (&)[4], int) {}
};
// ascending sort of 256 keys, 4 per lane (element e = register e / 64 of lane e % 64): 36 compare-exchange stages
__device__ __forceinline__ void bitonic_sort_256(uint32_t (&k)[4], int lane) {
  BitonicJ<2, 1>::run(k, lane);
  BitonicJ<4, 2>::run(k, lane);
  BitonicJ<8, 4>::run(k, lane);
  BitonicJ<16, 8>::run(k, lane);
  BitonicJ<32, 16>::run(k, lane);
  BitonicJ<64, 32>::run(k, lane);
  BitonicJ<128, 64>::run(k, lane);
  BitonicJ<256, 128>::run(k, lane);
}

// STATS: the development counters of ABD_GIBBS_STATS=1 (seven wave-uniform 64-bit counters: 14 scalar registers the product
// kernel does not have to spare)
template <typename R, bool STATS>
__global__ __launch_bounds__(64 * ABD_G2_MAX_WAVES, 3) void abd_gibbs_dense_kernel(const GibbsArgs ga) {
  extern __shared__ __align__(16) unsigned char smem[];
  const EvalArgs& a = ga.e;
  const int G = a.G, N = a.N, nt = a.nt;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + 2 * tstride;
  double* tab_e2 = reinterpret_cast<double*>(tab_ones + tstride);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_threads = (int)blockDim.x;  // 64 x the waves that fit the CU's LDS (abd_g2_waves)
  unsigned char* wb = reinterpret_cast<unsigned char*>(tab_e2 + ABD_EXP2_TAB) + (size_t)wave * abd_g2_wave_lds(G, (int)sizeof(R));
  YX<R>* dataN = reinterpret_cast<YX<R>*>(wb);
  YX<R>* dataS = reinterpret_cast<YX<R>*>(wb + abd_g2_pad16((size_t)G * sizeof(YX<R>)));
  double* suf = reinterpret_cast<double*>(wb + 2 * abd_g2_pad16((size_t)G * sizeof(YX<R>)));
  uint32_t* accw = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(suf) + abd_g2_pad16((size_t)(G + 1) * 8));
  uint16_t* plist = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(accw) + abd_g2_pad16((size_t)(G + 1) * 4));
  unsigned char* result = reinterpret_cast<unsigned char*>(plist) + abd_g2_pad16((size_t)(G + 1) * 2);

  const int c = blockIdx.y;  // one chain per block row
  const ChainPar& cp = a.ch[c];
  fill_pow_table(tabs, cp.rho_n, tstride, tid, n_threads);
  fill_pow_table(tabs + tstride, cp.rho_s, tstride, tid, n_threads);
  fill_ones_table(tab_ones, tstride, tid, n_threads);
  for (int e = tid; e < ABD_EXP2_TAB; e += n_threads) tab_e2[e] = a.exp2_tab[e];
  __syncthreads();

  // the chain's constants live in VECTOR registers: the scalar file is needed for the packed rows of the individual
  // (five rows of four words), and a spilled scalar costs a v_readlane in the walk
  G2Par p;
  p.perm_n = to_vgpr(cp.perm_n);
  p.temp_n = to_vgpr(cp.temp_n);
  p.rho_n = to_vgpr(cp.rho_n);
  p.init_n = to_vgpr(cp.init_n);
  p.perm_s = to_vgpr(cp.perm_s);
  p.rho_s = to_vgpr(cp.rho_s);
  p.init_s = to_vgpr(cp.init_s);
  p.c_n = to_vgpr(cp.b_n * (1.4426950408889634074 * ABD_EXP2_TAB));
  p.c_s = to_vgpr(cp.b_s * (1.4426950408889634074 * ABD_EXP2_TAB));
  p.d_n = to_vgpr(cp.d_n);
  p.d_s = to_vgpr(cp.d_s);
  p.nh_n = to_vgpr(-0.5 * ga.is2_n[c]);
  p.nh_s = to_vgpr(-0.5 * ga.is2_s[c]);
  const double theta0 = ga.theta0[c], theta7 = ga.theta7[c];
  // rho^(lane + 1) = table entry lane + 2 (only used when a previous round exists, i.e. G > 64 >= lane + 1)
  const double pwn = tabs[min(lane + 2, G)].x, pws_w = tabs[tstride + min(lane + 2, G)].x;
  const uint32_t k0 = ga.seed_lo ^ (ga.sweep * 0x9E3779B9u), k1 = ga.seed_hi;
  const uint32_t cs = ga.stream[c];
  uint64_t* rw = const_cast<uint64_t*>(cp.rw);
  int8_t* waner = const_cast<int8_t*>(cp.waner);
  uint64_t* iw = const_cast<uint64_t*>(cp.iw);
  int d_n1 = 0, d_m1 = 0;  // changes of sum(i_raw), sum(ab_s_waner) over this wave's individuals
  unsigned long long n_acc = 0, n_prop_total = 0;
  unsigned long long st_iter = 0, st_refill = 0, st_steps = 0, st_lane_steps = 0, st_tail = 0, st_commit = 0, st_inds = 0;

  for (;;) {
    // ---- next individual of this chain: one queue per chain, one individual per pop, so that the waves stay busy to the
    // end (guided chunks of up to 8 were measured: the pops themselves got cheaper -- a sweep that proposes nothing 0.54 ->
    // 0.28 ms -- but the coarser hand-out cost more at the end of a real sweep: 0.91 -> 0.95 ms converged, 2.26 -> 2.45 ms random)
    int j = 0;
    if (lane == 0) j = (int)atomicAdd(ga.work + c, 1u);
    j = __builtin_amdgcn_readfirstlane(j);
    if (j >= N) break;

    // ---- this individual's discrete state and data ----
    uint64_t V[ABD_MAXT], P[ABD_MAXT], Rw[ABD_MAXT], I[ABD_MAXT], I0[ABD_MAXT];
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) {
      V[t] = P[t] = Rw[t] = 0;
      if (t < nt) {
        V[t] = uniform_word(a.vw, (int64_t)t * N + j);
        if (a.pw) P[t] = uniform_word(a.pw, (int64_t)t * N + j);
        Rw[t] = uniform_word(rw, (int64_t)t * N + j);
        const int g = t * 64 + lane;
        if (g < G) {  // the individual's gap axis from the individual-major copy: contiguous, 1 KB per wave load
          dataN[g] = reinterpret_cast<const YX<R>*>(a.yxi_n)[(int64_t)j * G + g];
          dataS[g] = reinterpret_cast<const YX<R>*>(a.yxi_s)[(int64_t)j * G + g];
        }
      }
    }
    bool wj = __builtin_amdgcn_readfirstlane((int)waner[j]) != 0;
    const int firstV = first_bit(V);
    int pc0 = wj ? (1 << 16) : 0;  // sum(i_raw) and ab_s_waner of this individual before the sweep
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) pc0 += __builtin_popcountll(Rw[t]);

    // ---- random order of this individual's proposals ----
    // Philox words as abd_gibbs_kernel: word 0 orders the dims (low 9 bits = the dim), word 1 < 0.8 2^32 proposes the
    // dim, word 2 is the acceptance uniform.  Dims that are not proposed never enter the list.
    uint32_t key[4];
    int n_prop = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = r * 64 + lane;
      key[r] = 0xFFFFFFFFu;
      if (d < G) {
        const Philox4 rr = philox4x32_10((uint32_t)d, (uint32_t)j + ga.ind_offset, cs, 0u, k0, k1);
        if (rr.w[1] < ABD_TRANSIT_P_U32) {
          key[r] = (rr.w[0] & ~0x1FFu) | (uint32_t)d;
          accw[d] = rr.w[2];
        }
      }
      n_prop += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[r] != 0xFFFFFFFFu));
    }
    bitonic_sort_256(key, lane);
    // the ab_s_waner dim (dim G) takes its place among them
    const Philox4 rwz = philox4x32_10((uint32_t)G, (uint32_t)j + ga.ind_offset, cs, 0u, k0, k1);
    const bool w_proposed = rwz.w[1] < ABD_TRANSIT_P_U32;
    const uint32_t w_key = (rwz.w[0] & ~0x1FFu) | (uint32_t)G;
    int w_rank = 0;  // proposed i_raw dims ordered before it
#pragma unroll
    for (int r = 0; r < 4; ++r) w_rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[r] < w_key));
    if (!w_proposed) w_rank = 1 << 20;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = r * 64 + lane;
      if (key[r] != 0xFFFFFFFFu) plist[e + (e >= w_rank ? 1 : 0)] = (uint16_t)(key[r] & 0x1FFu);
    }
    if (w_proposed) {
      if (lane == 0) {
        plist[w_rank] = (uint16_t)G;
        accw[G] = rwz.w[2];
      }
      n_prop += 1;
    }
    for (int e = lane; e <= G; e += 64) result[e] = ABD_G2_PENDING;

    // ---- the current state: constrained infections, responses, terms, suffix sums ----
    double total_cur = 0.0;
    int firstI = 1 << 20;
    auto refresh = [&]() {
      constrain_i0(Rw, P, a, I0);
      const uint64_t none[ABD_MAXT] = {0, 0, 0, 0};
      three_gaps_from(I0, none, 0, I);
      firstI = first_bit(I);
      double un[ABD_MAXT], us[ABD_MAXT], term[ABD_MAXT];
      g2_eval_rounds<R>(a, p, lane, I, V, tabs, wj ? tabs + tstride : tab_ones, pwn, wj ? pws_w : 1.0, dataN, dataS, tab_e2, un,
                        us, term);
      double carry = 0.0;
      int ln = lane;  // (opaque: left to itself the compiler hoists the scan's six "lane + off < 64" masks out of the
      asm volatile("" : "+v"(ln));  // individual loop and keeps them in 12 scalar registers for the whole kernel)
#pragma unroll
      for (int t = ABD_MAXT - 1; t >= 0; --t) {
        if (t < nt) {
          const int g = t * 64 + lane;
          double x = -term[t];  // >= 0; lanes past the last gap hold 0
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const double y = __shfl_down(x, off, 64);
            x += ln + off < 64 ? y : 0.0;
          }
          x += carry;
          if (g < G) suf[g] = x;
          carry = readlane_f64(x, 0);
        }
      }
      if (lane == 0) suf[G] = 0.0;
      total_cur = -carry;
      __builtin_amdgcn_wave_barrier();
    };
    __builtin_amdgcn_wave_barrier();  // the data rows are in LDS
    refresh();

    // ---- the sweep ----
    int frontier = 0, next = 0;  // positions < frontier are committed; positions < next have been handed out
    // per-lane walk state
    bool active = false;
    int pidx = 0, g = 0, g_first = 0;
    double tn = 0.0, ts = 0.0, S = 0.0, B0 = 0.0, thr = 0.0, lu = 0.0;
    uint32_t cfn_hi = 0, cfs_hi = 0;
    uint64_t inw = 0, vw = 0, In[ABD_MAXT] = {0, 0, 0, 0};
    const uint32_t z_ei = zero_vgpr(), z_ev = zero_vgpr(), z_cn = zero_vgpr(), z_cs = zero_vgpr();

    bool dirty = false;  // a result has been written since the last commit scan
    if (STATS) ++st_inds;
    for (int iter = 0; iter < ABD_G2_ITER_CAP && frontier < n_prop; ++iter) {
      if (STATS) ++st_iter;
      // every iteration commits, hands out or advances at least one proposal, and an acceptance -- the only event that
      // moves `next` back -- changes the state for good: the loop ends; the cap only bounds a defect
      const double rho_j = wj ? p.rho_s : 1.0;  // abd.py:374
      // ---- 1. idle lanes pick up the next proposals ----
      const uint64_t idle_mask = __builtin_amdgcn_ballot_w64(!active);
      const int n_idle = __builtin_popcountll(idle_mask);
      if (next < n_prop && (n_idle >= ga.refill_min || n_idle == 64)) {
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0));
        const bool take = !active && next + rank < n_prop;
        if (STATS) ++st_refill;
        dirty = true;  // immediate results (no change / waning flip) may be among them
        if (take) {
          pidx = next + rank;
          const int d = plist[pidx];
          lu = log_uniform_u32(accw[d], tab_e2);  // log of the acceptance uniform, ~18 instructions
          if (d == G) {
            result[pidx] = ABD_G2_COMPLEX;  // ab_s_waner: evaluated by the whole wave at the frontier
          } else {
            // the flipped column's constraints: i0, then the three-gap pass from the first gap where i0 changes
            uint64_t Rn[ABD_MAXT], I0n[ABD_MAXT], X[ABD_MAXT], before[ABD_MAXT];
            const int dw = d >> 6;
            const uint64_t bit = 1ull << (d & 63);
            bool was_one = false;
#pragma unroll
            for (int t = 0; t < ABD_MAXT; ++t) {
              Rn[t] = Rw[t];
              if (t == dw) {
                was_one = (Rw[t] & bit) != 0;
                Rn[t] ^= bit;
              }
            }
            const double delta0 = was_one ? -theta0 : theta0;  // Bernoulli(i_raw | p) on the RAW matrix (abd.py:427)
            constrain_i0(Rn, P, a, I0n);
#pragma unroll
            for (int t = 0; t < ABD_MAXT; ++t) X[t] = I0n[t] ^ I0[t];
            int gf = 1 << 20;
            const int p0 = first_bit(X);
            if (p0 < (1 << 20)) {
#pragma unroll
              for (int t = 0; t < ABD_MAXT; ++t) {
                const int rel = p0 - t * 64;
                before[t] = I[t] & (rel <= 0 ? 0ull : (rel >= 64 ? ~0ull : ((1ull << rel) - 1ull)));
              }
              three_gaps_from(I0n, before, p0, In);
#pragma unroll
              for (int t = 0; t < ABD_MAXT; ++t) X[t] = In[t] ^ I[t];
              gf = first_bit(X);
            }
            if (gf >= (1 << 20)) {
              // the constrained infections do not change: the prior term decides
              result[pidx] = (delta0 > 0.0 || delta0 > lu) ? ABD_G2_ACCEPT : ABD_G2_REJECT;
            } else {
              B0 = delta0 + suf[gf];  // everything the current state holds from gf on is given up
              thr = lu - 1e-9 * (fabs(B0) + 1.0);
              // the two responses at gap gf - 1 of the CURRENT state (the states agree below gf): the dense design
              // (abd.py:258-274) summed over the current exposures, wave-uniform loops, per-lane table index
              {
                const double2_t* tsb = wj ? tabs + tstride : tab_ones;
                tn = ts = 0.0;
#pragma unroll
                for (int t = 0; t < ABD_MAXT; ++t) {
                  uint64_t m = I[t];
                  while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    const int idx = min(max(gf - (t * 64 + b), 0), G);  // 0 = "at or after gf": contributes nothing
                    tn += tabs[idx].x;
                    ts += tsb[idx].x;
                  }
                  m = V[t];
                  while (m) {
                    const int b = __builtin_ctzll(m);
                    m &= m - 1;
                    ts += tsb[min(max(gf - (t * 64 + b), 0), G)].x;
                  }
                }
              }
              cfn_hi = firstI < gf ? 0x3FF00000u : 0u;
              cfs_hi = min(firstI, firstV) < gf ? 0x3FF00000u : 0u;
              inw = word_at(In, gf) >> (gf & 63);
              vw = word_at(V, gf) >> (gf & 63);
              S = 0.0;
              g = g_first = gf;
              active = true;
            }
          }
        }
        next = min(n_prop, next + n_idle);
      }

      // ---- 2. one gap for every walking lane ----
      bool finished = false;
      if (active) {
        const uint32_t ei_hi = (uint32_t)__builtin_amdgcn_sbfe((int)(uint32_t)inw, 0u, 1u) & 0x3FF00000u;
        const uint32_t ev_hi = (uint32_t)__builtin_amdgcn_sbfe((int)(uint32_t)vw, 0u, 1u) & 0x3FF00000u;
        inw >>= 1;
        vw >>= 1;
        const double e_i = hi_to_double(ei_hi, z_ei), e_v = hi_to_double(ev_hi, z_ev);
        cfn_hi |= ei_hi;
        cfs_hi |= ei_hi | ev_hi;
        tn = fma(p.rho_n, tn, e_i);
        ts = fma(rho_j, ts, e_i + e_v);  // unit boosts: temp unused (abd.py:272)
        const double an = fma(p.temp_n, tn, fma(hi_to_double(cfn_hi, z_cn), p.perm_n, p.init_n));
        const double as = fma(hi_to_double(cfs_hi, z_cs), p.perm_s, p.init_s) + ts;
        const YX<R> on = dataN[g], os = dataS[g];
        S += g2_term(an, (double)on.x, (double)on.y, p.c_n, p.d_n, p.nh_n, as, (double)os.x, (double)os.y, p.c_s, p.d_s, p.nh_s,
                     tab_e2);
        ++g;
        // every remaining term is <= 0: delta <= B0 + S; the factor keeps the test on the safe side of rounding
        const bool dead = fma(S, 1.0 - 1e-9, B0) < thr;
        if (dead || g >= G) {
          const double delta = B0 + S;
          result[pidx] = (!dead && (delta > 0.0 || delta > lu)) ? ABD_G2_ACCEPT : ABD_G2_REJECT;
          active = false;
          finished = true;
        } else if ((g & 63) == 0) {
          inw = word_at(In, g);
          vw = word_at(V, g);
        }
      }
      dirty |= __builtin_amdgcn_ballot_w64(finished) != 0;
      // ---- 2b. the tail of an individual: every proposal has been handed out and only a few lanes still walk (a walk
      // that is not rejected early is as long as the gaps that are left).  The whole wave finishes one of them per
      // iteration, lanes = the remaining gaps: the bound at the last gap IS delta.
      const uint64_t walking = __builtin_amdgcn_ballot_w64(active);
      if (STATS) {
        const uint64_t stepped = walking | __builtin_amdgcn_ballot_w64(finished);
        st_steps += stepped != 0;
        st_lane_steps += (unsigned long long)__builtin_popcountll(stepped);
      }
      // (only walks that have already survived tail_age gaps: a young one is most likely rejected within a few more)
      const uint64_t old_walkers = __builtin_amdgcn_ballot_w64(active && g - g_first >= ga.tail_age);
      if (next >= n_prop && old_walkers != 0 && __builtin_popcountll(walking) <= ga.tail_lanes) {
        const int L = __builtin_ctzll(old_walkers);
        if (STATS) ++st_tail;
        const int g_l = __builtin_amdgcn_readlane(g, L);
        uint64_t In_l[ABD_MAXT], Is[ABD_MAXT], Vs[ABD_MAXT];
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t)
          In_l[t] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(In[t] >> 32), L) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)In[t], L);
        shift_row_down(In_l, g_l, Is);
        shift_row_down(V, g_l, Vs);
        const bool ci0 = __builtin_amdgcn_readlane((int)cfn_hi, L) != 0, civ0 = __builtin_amdgcn_readlane((int)cfs_hi, L) != 0;
        double un[ABD_MAXT], us[ABD_MAXT], term[ABD_MAXT];
        g2_eval_rounds<R>(a, p, lane, Is, Vs, tabs, wj ? tabs + tstride : tab_ones, pwn, wj ? pws_w : 1.0, dataN, dataS, tab_e2, un,
                          us, term, g_l, readlane_f64(tn, L), readlane_f64(ts, L), ci0, civ0);
        double tsum = 0.0;
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t) tsum += term[t];
        const double rest = wave_sum_uniform(tsum);
        if (lane == L) {
          const double delta = B0 + (S + rest);
          result[pidx] = (delta > 0.0 || delta > lu) ? ABD_G2_ACCEPT : ABD_G2_REJECT;
          active = false;
        }
        dirty = true;
      }
      __builtin_amdgcn_wave_barrier();

      // ---- 3. commit in the sweep's order ----
      if (!dirty) continue;
      dirty = false;
      if (STATS) ++st_commit;
      for (;;) {
        const int pos = frontier + lane;
        const int r = pos < next ? (int)result[pos] : ABD_G2_PENDING;
        const uint64_t stop = __builtin_amdgcn_ballot_w64(r != ABD_G2_REJECT);
        const int n_rej = stop ? __builtin_ctzll(stop) : 64;
        frontier += n_rej;
        if (n_rej == 64 && frontier < next) continue;  // a whole window of rejections: look further
        if (frontier >= next) break;
        const int rr = __builtin_amdgcn_readlane(r, n_rej & 63);
        if (rr == ABD_G2_PENDING) break;
        const int d = __builtin_amdgcn_readfirstlane((int)plist[frontier]);
        bool accepted = rr == ABD_G2_ACCEPT;
        if (rr == ABD_G2_COMPLEX) {
          // the waning flip changes rho_j at every gap: the whole wave evaluates the proposed state
          const bool wn = !wj;
          double un[ABD_MAXT], us[ABD_MAXT], term[ABD_MAXT];
          g2_eval_rounds<R>(a, p, lane, I, V, tabs, wn ? tabs + tstride : tab_ones, pwn, wn ? pws_w : 1.0, dataN, dataS, tab_e2,
                            un, us, term);
          double tsum = 0.0;
#pragma unroll
          for (int t = 0; t < ABD_MAXT; ++t) tsum += term[t];
          const double delta = (wn ? theta7 : -theta7) + (wave_sum_uniform(tsum) - total_cur);  // Bernoulli(waner | p_waner) abd.py:373
          const double log_u = readfirstlane_f64(log_uniform_u32(accw[G], tab_e2));
          accepted = delta > 0.0 || delta > log_u;
          if (accepted) wj = wn;
        } else if (accepted) {
#pragma unroll
          for (int t = 0; t < ABD_MAXT; ++t)
            if (t == (d >> 6)) Rw[t] ^= 1ull << (d & 63);
        }
        frontier += 1;
        if (!accepted) continue;
        // the state has changed: everything evaluated beyond this proposal is void
        ++n_acc;
        for (int e = frontier + lane; e < next; e += 64) result[e] = ABD_G2_PENDING;
        next = frontier;
        active = false;
        refresh();
        break;
      }
    }
    n_prop_total += (unsigned long long)frontier;

    // ---- write the individual's state back: raw bits, waning flag, and what the slot keeps beside them (the
    // constrained words the evaluation kernels read, the changes of sum(i_raw) and sum(ab_s_waner)) ----
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t)
        if (t < nt) {
          rw[(int64_t)t * N + j] = Rw[t];
          iw[(int64_t)t * N + j] = I[t];
        }
      waner[j] = wj ? 1 : 0;
    }
    int pc1 = wj ? (1 << 16) : 0;
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) pc1 += __builtin_popcountll(Rw[t]);
    d_n1 += (pc1 & 0xFFFF) - (pc0 & 0xFFFF);
    d_m1 += (pc1 >> 16) - (pc0 >> 16);
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0 && (d_n1 | d_m1)) {
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(const_cast<long long*>(cp.cnt));
    atomicAdd(cnt + 0, (unsigned long long)(long long)d_n1);  // two's complement: a negative change wraps to the right sum
    atomicAdd(cnt + 1, (unsigned long long)(long long)d_m1);
  }
  if (lane == 0 && (n_acc | n_prop_total)) {
    atomicAdd(ga.counts + 2 * c + 0, n_acc);
    atomicAdd(ga.counts + 2 * c + 1, n_prop_total);
  }
  if (STATS && lane == 0 && ga.stats) {
    atomicAdd(ga.stats + 0, st_inds);
    atomicAdd(ga.stats + 1, st_iter);
    atomicAdd(ga.stats + 2, st_refill);
    atomicAdd(ga.stats + 3, st_steps);
    atomicAdd(ga.stats + 4, st_lane_steps);
    atomicAdd(ga.stats + 5, st_tail);
    atomicAdd(ga.stats + 6, st_commit);
    atomicAdd(ga.stats + 7, n_acc);
  }
}
