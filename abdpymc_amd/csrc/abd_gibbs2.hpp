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
//                           SPECULATIVELY against the current state, each by its own lane: the lane works out how the
//                           flip changes the constrained infections (abd.py:640-667), then walks gap by gap from the
//                           first gap whose constrained infection changes, carrying the two responses by the
//                           recurrence (abd.py:288) and adding up the new likelihood terms.  Every term is <= 0, so
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
// The individual's state lives in LDS as what a proposal needs of it, not as packed rows in scalar registers (rounds 2 and 3:
// five rows of 4 -- or 8 -- 64-bit words held wave-uniform filled the scalar file: 210 spilled registers, a scratch array for
// the per-lane row index, and no room for cohorts beyond 256 gaps):
//   * i0 (the infections before the three-gap pass: abd.py:643-647, or per time chunk abd.py:818 + 771), the kept
//     infections I and the vaccinations as SORTED POSITION LISTS -- an individual has a handful of each;
//   * per time chunk: does PCR+ decide it, and the first two raw infections in it.  A flip of raw bit d then changes i0 by
//     at most one position leaving and one entering, worked out from those few numbers;
//   * the three-gap pass restarted at the first changed position is a walk over the i0 list (abd.py:560-601: a set bit is
//     kept iff no kept bit lies in the three gaps before it), compared entry by entry with the current I list: the first
//     difference is the gap the lane's walk starts at, and the new kept infections from there on are the lane's own short
//     list (ABD_G2_KCAP entries; a proposal that needs more -- e.g. on an all-ones i_raw -- is evaluated by the whole wave at
//     the frontier, like the waning flip);
//   * the walk meets infections and vaccinations by comparing its gap with the next list entry.
// The packed rows themselves are kept in LDS too (the raw row for the flip's old value, all rows for the whole-wave
// evaluations and the write-back) and only pass through scalar registers where a whole-wave evaluation runs.  Nothing here
// depends on the number of words: the kernel is a template on it (4: <= 256 gaps, 8: <= 512) only for those evaluations.
//
// Per wave in LDS: the individual's OD pairs of both antigens, the suffix sums of the current terms, log u per dim, the
// proposal list, one result byte per proposal, the rows, the position lists and the lanes' new-infection lists.
#pragma once

#include "abd_gibbs.hpp"

#define ABD_G2_PENDING 0
#define ABD_G2_REJECT 1
#define ABD_G2_ACCEPT 2
#define ABD_G2_COMPLEX 3
#define ABD_G2_ITER_CAP (1 << 20)    // hard bound on scheduler iterations per individual (never reached: see the loop)
#define ABD_G2_KCAP 8                // new kept infections (from the first changed gap on) a lane holds for its walk
#define ABD_G2_NONE (1 << 20)        // "no position"
#ifndef ABD_G2_STEPS
#define ABD_G2_STEPS 6               // gaps a walking lane takes per scheduler iteration at most
#endif

#define ABD_G2_MAX_WAVES 12  // waves of a workgroup (= of a CU: one workgroup per CU, three waves per SIMD, <= 168 registers)
#define ABD_G2_MAX_WAVES_WIDE 8  // ... of the 512-gap kernel: its LDS holds fewer anyway, and two waves per SIMD may use 256 registers
__host__ __device__ constexpr size_t abd_g2_pad16(size_t b) { return (b + 15) / 16 * 16; }
__host__ __device__ constexpr int abd_g2_words(int G) { return (G + 63) / 64 > ABD_MAXT ? ABD_MAXT_MAX : ABD_MAXT; }
// per-wave LDS bytes (12.4 KB at G = 200, fp64: LDS, not registers, decides how many waves a CU holds)
struct G2Layout {
  size_t dataS, suf, accw, plist, result, rows, epos, i0pos, ipos, tpos, vpos, inl, total;
};
__host__ __device__ constexpr G2Layout abd_g2_layout(int G, int rbytes) {
  G2Layout L{};
  size_t b = abd_g2_pad16((size_t)G * 2 * rbytes);  // {od, log_dilution} per gap, N ...
  L.dataS = b;
  b += abd_g2_pad16((size_t)G * 2 * rbytes);         // ... then S
  L.suf = b;
  b += abd_g2_pad16((size_t)(G + 1) * 8);            // suf[g] = -(sum of the current terms of gaps >= g); suf[G] = 0
  L.accw = b;
  b += abd_g2_pad16((size_t)(G + 1) * 4);            // the acceptance draw's Philox word by dim
  L.plist = b;
  b += abd_g2_pad16((size_t)(G + 1) * 2);            // proposal list: dim by position in the sweep
  L.result = b;
  b += abd_g2_pad16((size_t)(G + 1));                // result by position
  L.rows = b;
  b += abd_g2_pad16((size_t)4 * abd_g2_words(G) * 8);  // packed rows: vaccinations, PCR+, i_raw, kept infections I
  L.epos = b;
  b += abd_g2_pad16((size_t)2 * (G + 1) * 2);        // exposures in the order their responses are summed: per 64-gap word the kept infections, then the vaccinations (bit 15: a vaccination)
  L.i0pos = b;
  b += abd_g2_pad16((size_t)(G + 1) * 2);            // positions of i0, ascending
  L.ipos = b;
  b += abd_g2_pad16((size_t)(G / 4 + 2) * 2);        // positions of the kept infections I, ascending (at most one in four gaps)
  L.tpos = b;
  b += abd_g2_pad16((size_t)(G / 4 + 2) * 2);        // the kept infections of a proposed state the whole wave evaluates (ABD_G2_COMPLEX)
  L.vpos = b;
  b += abd_g2_pad16((size_t)(G + 1) * 2);            // positions of the vaccinations, ascending
  L.inl = b;
  b += (size_t)ABD_G2_KCAP * 64 * 2;                 // [ABD_G2_KCAP][64] a lane's new kept infections from its first changed gap on
  L.total = b;
  return L;
}
// (the kernel lays a wave's regions out for the gap CAPACITY of its template, 256 or 512, so that their offsets are immediates)
__host__ __device__ inline size_t abd_g2_wave_lds(int G, int rbytes) { return abd_g2_layout(G, rbytes).total; }
// LDS of the tables every wave of the workgroup shares: [2][G+1] power tables + [G+1] ones + 2^(j/1024) + the chunk masks
__host__ __device__ inline size_t abd_g2_shared_lds(int G) {
  return (size_t)3 * (G + 1) * sizeof(double2_t) + (size_t)ABD_EXP2_TAB * sizeof(double) + (size_t)3 * ABD_MAXT_MAX * 8;
}
// waves per workgroup that fit the CU's 160 KB (0: not even one)
__host__ __device__ inline int abd_g2_waves(int G, int rbytes) {
  const size_t avail = (size_t)160 * 1024 - abd_g2_shared_lds(G);
  const size_t w = avail / abd_g2_wave_lds(G, rbytes);
  const size_t cap = abd_g2_words(G) > ABD_MAXT ? ABD_G2_MAX_WAVES_WIDE : ABD_G2_MAX_WAVES;
  return (int)(w > cap ? cap : w);
}
__host__ __device__ inline size_t abd_g2_lds(int G, int rbytes, int n_waves) {
  return abd_g2_shared_lds(G) + (size_t)n_waves * abd_g2_wave_lds(G, rbytes);
}

// i0 of constrain_infections before the three-gap pass (abd.py:643-647 one chunk; abd.py:818 + 771 per chunk otherwise).
// Works on wave-uniform and on per-lane words alike.
// cmk: the chunk masks, [3][ABD_MAXT_MAX] (EvalArgs::chunk_mask; the sweep kernel keeps a copy in LDS)
template <int MT>
__device__ __forceinline__ void constrain_i0(const uint64_t (&raw)[MT], const uint64_t (&pcr)[MT], int n_chunks, const uint64_t* cmk,
                                             uint64_t (&i0)[MT]) {
  if (n_chunks <= 1) {
#pragma unroll
    for (int t = 0; t < MT; ++t) i0[t] = raw[t] | pcr[t];
  } else {
#pragma unroll
    for (int t = 0; t < MT; ++t) i0[t] = 0;
    for (int c = 0; c < n_chunks; ++c) {
      bool has_pcr = false;
#pragma unroll
      for (int t = 0; t < MT; ++t) has_pcr |= (pcr[t] & cmk[c * ABD_MAXT_MAX + t]) != 0;
      bool found = false;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const uint64_t cm = cmk[c * ABD_MAXT_MAX + t];
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
template <int MT>
__device__ __forceinline__ void three_gaps_from(const uint64_t (&i0)[MT], const uint64_t (&before)[MT], int p0,
                                                uint64_t (&out)[MT]) {
  int block_until = 0;
#pragma unroll
  for (int t = MT - 1; t >= 0; --t)
    if (before[t] != 0 && block_until == 0) block_until = t * 64 + 63 - __builtin_clzll(before[t]) + 4;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
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

template <int MT>
__device__ __forceinline__ int first_bit(const uint64_t (&w)[MT]) {  // position of the lowest set bit, or 1 << 20
  int p = 1 << 20;
#pragma unroll
  for (int t = MT - 1; t >= 0; --t)
    if (w[t] != 0) p = t * 64 + __builtin_ctzll(w[t]);
  return p;
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
// The state comes as sorted position lists in LDS (no packed rows in scalar registers): infections il[k * istride], k in
// [ki, ni), vaccinations vl[k], k in [kv, nv), all at or after g_off.  The rounds start at gap g_off (0: the whole individual;
// otherwise cvn / cvs / ci / civ are the state at gap g_off - 1: the rest of one lane's walk, taken over by the whole wave).
// Within a round the infections are added in ascending order, then the vaccinations -- the order of the packed rows' words.
template <typename R, int MT>
__device__ __forceinline__ void g2_eval_rounds(int G, const G2Par& p, int lane, const uint16_t* il, int istride, int ki, int ni,
                                               const uint16_t* vl, int kv, int nv, const double2_t* tab_n, const double2_t* tab_s,
                                               double pwn, double pws, const YX<R>* dataN, const YX<R>* dataS,
                                               const double* tab_e2, double (&term_o)[MT], int g_off = 0, double cvn = 0.0,
                                               double cvs = 0.0, bool ci = false, bool civ = false) {
  // cvn / cvs: responses at the end of the previous round (wave-uniform)
  const int n_rounds = (G - g_off + 63) >> 6;
  // the first listed infection / exposure: from there on the permanent responses are switched on (abd.py:306)
  const int first_i = ki < ni ? __builtin_amdgcn_readfirstlane((int)il[ki * istride]) : ABD_G2_NONE;
  const int first_v = kv < nv ? __builtin_amdgcn_readfirstlane((int)vl[kv]) : ABD_G2_NONE;
  const int first_iv = min(first_i, first_v);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    term_o[t] = 0.0;
    if (t < n_rounds) {
      const int r0 = g_off + t * 64, r1 = r0 + 64;  // the round's gaps
      double un = pwn * cvn, us = pws * cvs;
      while (ki < ni) {  // wave-uniform loop over this round's infections
        const int pos = __builtin_amdgcn_readfirstlane((int)il[ki * istride]);
        if (pos >= r1) break;
        const int idx = min(max(lane - (pos - r0) + 1, 0), G);  // 0 = "in the future"
        un += tab_n[idx].x;
        us += tab_s[idx].x;
        ++ki;
      }
      while (kv < nv) {
        const int pos = __builtin_amdgcn_readfirstlane((int)vl[kv]);
        if (pos >= r1) break;
        us += tab_s[min(max(lane - (pos - r0) + 1, 0), G)].x;
        ++kv;
      }
      const int g = r0 + lane;
      const bool cum_i = ci || first_i <= g;
      const bool cum_iv = civ || first_iv <= g;
      const bool valid = g < G;
      const int gg = valid ? g : 0;
      const YX<R> on = dataN[gg], os = dataS[gg];
      const double an = p.init_n + (cum_i ? p.perm_n : 0.0) + p.temp_n * un;
      const double as = p.init_s + (cum_iv ? p.perm_s : 0.0) + us;
      const double term = g2_term(an, (double)on.x, (double)on.y, p.c_n, p.d_n, p.nh_n, as, (double)os.x, (double)os.y, p.c_s,
                                  p.d_s, p.nh_s, tab_e2);
      term_o[t] = valid ? term : 0.0;
      cvn = readlane_f64(un, 63);
      cvs = readlane_f64(us, 63);
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
// element e = r * 64 + lane of NR x 64 keys; direction of its sub-sequence: ascending iff (e & K) == 0 (the last merge: all)
template <int NR, int K, int J>
__device__ __forceinline__ void bitonic_stage(uint32_t (&k)[NR], int lane) {
  if (J >= 64) {
    constexpr int dr = J / 64;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if ((r & dr) == 0) {
        const bool asc = K >= NR * 64 ? true : ((r * 64) & K) == 0;
        const uint32_t lo = min(k[r], k[r | dr]), hi = max(k[r], k[r | dr]);
        k[r] = asc ? lo : hi;
        k[r | dr] = asc ? hi : lo;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const bool asc = K >= NR * 64 ? true : (K >= 64 ? ((r * 64) & K) == 0 : (lane & K) == 0);
      k[r] = bitonic_lane_step<(J < 64 ? J : 1)>(k[r], lane, asc);
    }
  }
}
template <int NR, int K, int J>
struct BitonicJ {
  static __device__ __forceinline__ void run(uint32_t (&k)[NR], int lane) {
    bitonic_stage<NR, K, J>(k, lane);
    BitonicJ<NR, K, J / 2>::run(k, lane);
  }
};
template <int NR, int K>
struct BitonicJ<NR, K, 0> {
  static __device__ __forceinline__ void run(uint32_t (&)[NR], int) {}
};
template <int NR, int K>
struct BitonicK {
  static __device__ __forceinline__ void run(uint32_t (&k)[NR], int lane) {
    BitonicK<NR, K / 2>::run(k, lane);
    BitonicJ<NR, K, K / 2>::run(k, lane);
  }
};
template <int NR>
struct BitonicK<NR, 1> {
  static __device__ __forceinline__ void run(uint32_t (&)[NR], int) {}
};
// ascending sort of NR x 64 keys, NR per lane (element e = register e / 64 of lane e % 64): 36 compare-exchange stages for
// 256 keys, 45 for 512
template <int NR>
__device__ __forceinline__ void bitonic_sort(uint32_t (&k)[NR], int lane) {
  BitonicK<NR, NR * 64>::run(k, lane);
}

// a packed row of the wave's individual from LDS into scalar registers
template <int MT>
__device__ __forceinline__ void g2_load_row(const uint64_t* row /* LDS */, uint64_t (&w)[MT]) {
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const uint64_t v = row[t];
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    w[t] = ((uint64_t)hi << 32) | lo;
  }
}
// positions of the set bits of word t of a wave-uniform row, ascending, appended to list[n ..) (LDS; lanes = bits); returns the new count
__device__ __forceinline__ int g2_append_positions(uint64_t m, int t, int lane, uint16_t* list, int n, uint16_t flag) {
  if (m != 0) {  // (wave-uniform)
    if ((m >> lane) & 1ull) list[n + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)(t * 64 + lane) | flag;
    n += __builtin_popcountll(m);
  }
  return n;
}

// STATS: the development counters of ABD_GIBBS_STATS=1 (seven wave-uniform 64-bit counters)
template <typename R, bool STATS, int MT>
__global__ __launch_bounds__(64 * (MT > ABD_MAXT ? ABD_G2_MAX_WAVES_WIDE : ABD_G2_MAX_WAVES), MT > ABD_MAXT ? 2 : 3) void abd_gibbs_dense_kernel(const GibbsArgs ga) {
  extern __shared__ __align__(16) unsigned char smem[];
  const EvalArgs& a = ga.e;
  const int G0 = a.G, N = a.N, nt0 = a.nt, nch0 = a.n_chunks;
  const int G = G0;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + 2 * tstride;
  double* tab_e2 = reinterpret_cast<double*>(tab_ones + tstride);
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_threads = (int)blockDim.x;  // 64 x the waves that fit the CU's LDS (abd_g2_waves)
  const G2Layout L = abd_g2_layout(G0, (int)sizeof(R));
  // the chunk masks, [3][ABD_MAXT_MAX], are read from LDS where they are needed: as kernel arguments the compiler loaded all 24
  // words in front of the individual loop and kept them, spilled, for the whole kernel
  uint64_t* cmk = reinterpret_cast<uint64_t*>(tab_e2 + ABD_EXP2_TAB);
  unsigned char* wb = reinterpret_cast<unsigned char*>(cmk + 3 * ABD_MAXT_MAX) + (size_t)wave * L.total;
  YX<R>* dataN = reinterpret_cast<YX<R>*>(wb);
  YX<R>* dataS = reinterpret_cast<YX<R>*>(wb + L.dataS);
  double* suf = reinterpret_cast<double*>(wb + L.suf);
  uint32_t* accw = reinterpret_cast<uint32_t*>(wb + L.accw);
  uint16_t* plist = reinterpret_cast<uint16_t*>(wb + L.plist);
  unsigned char* result = wb + L.result;
  uint64_t* row_v = reinterpret_cast<uint64_t*>(wb + L.rows);  // packed rows of the individual: vaccinations,
  uint64_t* row_p = row_v + MT;                                 // PCR+,
  uint64_t* row_r = row_p + MT;                                 // i_raw,
  uint64_t* row_i = row_r + MT;                                 // the kept infections I = constrain(i_raw, pcrpos)
  uint16_t* epos = reinterpret_cast<uint16_t*>(wb + L.epos);
  uint16_t* i0pos = reinterpret_cast<uint16_t*>(wb + L.i0pos);
  uint16_t* ipos = reinterpret_cast<uint16_t*>(wb + L.ipos);
  uint16_t* tpos = reinterpret_cast<uint16_t*>(wb + L.tpos);
  uint16_t* vpos = reinterpret_cast<uint16_t*>(wb + L.vpos);
  uint16_t* inl = reinterpret_cast<uint16_t*>(wb + L.inl) + lane0;  // this lane's list: entry k at inl[k * 64]

  const int c = blockIdx.y;  // one chain per block row
  const ChainPar& cp = a.ch[c];
  fill_pow_table(tabs, cp.rho_n, tstride, tid, n_threads);
  fill_pow_table(tabs + tstride, cp.rho_s, tstride, tid, n_threads);
  fill_ones_table(tab_ones, tstride, tid, n_threads);
  for (int e = tid; e < ABD_EXP2_TAB; e += n_threads) tab_e2[e] = a.exp2_tab[e];
  if (tid < 3 * ABD_MAXT_MAX) cmk[tid] = a.chunk_mask[tid / ABD_MAXT_MAX][tid % ABD_MAXT_MAX];
  __syncthreads();

  // the chain's constants live in VECTOR registers (a spilled scalar costs a v_readlane in the walk)
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
  const double pwn = tabs[min(lane0 + 2, G)].x, pws_w = tabs[tstride + min(lane0 + 2, G)].x;
  const uint32_t k0_0 = ga.seed_lo ^ (ga.sweep * 0x9E3779B9u), k1_0 = ga.seed_hi;
  const uint32_t cs = ga.stream[c];
  uint64_t* rw = const_cast<uint64_t*>(cp.rw);
  int8_t* waner = const_cast<int8_t*>(cp.waner);
  uint64_t* iw = const_cast<uint64_t*>(cp.iw);
  // where the time chunks begin (abd.py:865-882; an empty chunk begins nowhere)
  int chunk_lo1 = ABD_G2_NONE, chunk_lo2 = ABD_G2_NONE;
  if (a.n_chunks > 1) {
    uint64_t cm[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) cm[t] = a.chunk_mask[1][t];
    chunk_lo1 = first_bit<MT>(cm);
    if (a.n_chunks > 2) {
#pragma unroll
      for (int t = 0; t < MT; ++t) cm[t] = a.chunk_mask[2][t];
      chunk_lo2 = first_bit<MT>(cm);
    }
  }
  int d_n1 = 0, d_m1 = 0;  // changes of sum(i_raw), sum(ab_s_waner) over this wave's individuals
  unsigned int n_acc = 0, n_prop_total = 0;
  unsigned long long st_iter = 0, st_refill = 0, st_steps = 0, st_lane_steps = 0, st_tail = 0, st_commit = 0, st_inds = 0;

  for (;;) {
    // (the lane index of this iteration is opaque to the compiler: left to itself it hoists every lane predicate of the loop --
    // the 36 stages of the sort alone have two dozen, "(lane & J) == 0" -- out of it and keeps them as 64-bit masks for the
    // whole kernel: 90 of the 287 scalar registers it then had to spill, reloaded at every use)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    // (likewise the word and chunk counts: their comparisons with 0 .. MT - 1 were kept as nineteen 64-bit select masks)
    // and the twenty round keys of the Philox stream, which only the set-up of an individual needs)
    const int nt = gibbs_opaque_uniform(nt0), nch = gibbs_opaque_uniform(nch0), G = gibbs_opaque_uniform(G0);
    const uint32_t k0 = (uint32_t)gibbs_opaque_uniform((int)k0_0), k1 = (uint32_t)gibbs_opaque_uniform((int)k1_0);
    // ---- next individual of this chain: one queue per chain, one individual per pop, so that the waves stay busy to the
    // end (guided chunks of up to 8 were measured: the pops themselves got cheaper -- a sweep that proposes nothing 0.54 ->
    // 0.28 ms -- but the coarser hand-out cost more at the end of a real sweep: 0.91 -> 0.95 ms converged, 2.26 -> 2.45 ms random)
    int j = 0;
    if (lane == 0) j = (int)atomicAdd(ga.work + c, 1u);
    j = __builtin_amdgcn_readfirstlane(j);
    if (j >= N) break;

    // ---- this individual's discrete state and data: into LDS ----
    int firstV = ABD_G2_NONE, n_v = 0, pc0 = 0;
    {
      uint64_t V[MT], P[MT], Rw[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        V[t] = P[t] = Rw[t] = 0;
        if (t < nt) {
          V[t] = uniform_word(a.vw, (int64_t)t * N + j);
          if (a.pw) P[t] = uniform_word(a.pw, (int64_t)t * N + j);
          Rw[t] = uniform_word(rw, (int64_t)t * N + j);
        }
      }
      for (int g = lane; g < G; g += 64) {  // the individual's gap axis from the individual-major copy: contiguous, 1 KB per wave load
        dataN[g] = reinterpret_cast<const YX<R>*>(a.yxi_n)[(int64_t)j * G + g];
        dataS[g] = reinterpret_cast<const YX<R>*>(a.yxi_s)[(int64_t)j * G + g];
      }
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          row_v[t] = V[t];
          row_p[t] = P[t];
          row_r[t] = Rw[t];
        }
      }
      firstV = first_bit<MT>(V);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        n_v = g2_append_positions(V[t], t, lane, vpos, n_v, 0);
        pc0 += __builtin_popcountll(Rw[t]);
      }
    }
    bool wj = __builtin_amdgcn_readfirstlane((int)waner[j]) != 0;
    pc0 += wj ? (1 << 16) : 0;  // sum(i_raw) and ab_s_waner of this individual before the sweep

    // ---- random order of this individual's proposals ----
    // Philox words as abd_gibbs_kernel: word 0 orders the dims (low 9 bits = the dim), word 1 < 0.8 2^32 proposes the
    // dim, word 2 is the acceptance uniform.  Dims that are not proposed never enter the list.
    int n_prop = 0, w_rank = 0;
    bool w_proposed;
    {
      uint32_t key[MT];
#pragma unroll
      for (int r = 0; r < MT; ++r) {
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
      bitonic_sort<MT>(key, lane);
      // the ab_s_waner dim (dim G) takes its place among them
      const Philox4 rwz = philox4x32_10((uint32_t)G, (uint32_t)j + ga.ind_offset, cs, 0u, k0, k1);
      w_proposed = rwz.w[1] < ABD_TRANSIT_P_U32;
      const uint32_t w_key = (rwz.w[0] & ~0x1FFu) | (uint32_t)G;
#pragma unroll
      for (int r = 0; r < MT; ++r) w_rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[r] < w_key));  // proposed i_raw dims ordered before it
      if (!w_proposed) w_rank = 1 << 20;
#pragma unroll
      for (int r = 0; r < MT; ++r) {
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
    }
    for (int e = lane; e <= G; e += 64) result[e] = ABD_G2_PENDING;

    // ---- the current state: constrained infections, position lists, chunk facts, terms, suffix sums ----
    double total_cur = 0.0;
    int firstI = ABD_G2_NONE, n_i = 0, n_i0 = 0, n_e = 0;
    // per time chunk (several chunks only): bit c of chunk_pcr = PCR+ decides chunk c; the first two raw infections in it
    int chunk_pcr = 0, f1_0 = ABD_G2_NONE, f1_1 = ABD_G2_NONE, f1_2 = ABD_G2_NONE, f2_0 = ABD_G2_NONE, f2_1 = ABD_G2_NONE, f2_2 = ABD_G2_NONE;
    auto refresh = [&]() {
      uint64_t V[MT], I[MT];
      {
        uint64_t Rw[MT], P[MT], I0[MT];
        g2_load_row<MT>(row_r, Rw);
        g2_load_row<MT>(row_p, P);
        g2_load_row<MT>(row_v, V);
        constrain_i0<MT>(Rw, P, nch, cmk, I0);
        uint64_t none[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) none[t] = 0;
        three_gaps_from<MT>(I0, none, 0, I);
        n_i = n_i0 = n_e = 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          n_i0 = g2_append_positions(I0[t], t, lane, i0pos, n_i0, 0);
          n_i = g2_append_positions(I[t], t, lane, ipos, n_i, 0);
          n_e = g2_append_positions(I[t], t, lane, epos, n_e, 0);
          n_e = g2_append_positions(V[t], t, lane, epos, n_e, 0x8000u);
        }
        if (nch > 1) {
          chunk_pcr = 0;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            int f1 = ABD_G2_NONE, f2 = ABD_G2_NONE;
            if (cc < nch) {
              bool has = false;
              uint64_t r[MT];
#pragma unroll
              for (int t = 0; t < MT; ++t) {
                const uint64_t cm = cmk[cc * ABD_MAXT_MAX + t];
                has |= (P[t] & cm) != 0;
                r[t] = Rw[t] & cm;
              }
              if (has) chunk_pcr |= 1 << cc;
              f1 = first_bit<MT>(r);
              if (f1 < ABD_G2_NONE) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                  if (t == (f1 >> 6)) r[t] &= r[t] - 1;  // (the lowest set bit of the row lies in this word)
                f2 = first_bit<MT>(r);
              }
            }
            if (cc == 0) f1_0 = f1, f2_0 = f2;
            if (cc == 1) f1_1 = f1, f2_1 = f2;
            if (cc == 2) f1_2 = f1, f2_2 = f2;
          }
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) row_i[t] = I[t];
      }
      firstI = first_bit<MT>(I);
      double term[MT];
      __builtin_amdgcn_wave_barrier();  // (the lists are in LDS)
      g2_eval_rounds<R, MT>(G, p, lane, ipos, 1, 0, n_i, vpos, 0, n_v, tabs, wj ? tabs + tstride : tab_ones, pwn, wj ? pws_w : 1.0, dataN, dataS,
                            tab_e2, term);
      double carry = 0.0;
      int ln = lane;  // (opaque: left to itself the compiler hoists the scan's six "lane + off < 64" masks out of the
      asm volatile("" : "+v"(ln));  // individual loop and keeps them in 12 scalar registers for the whole kernel)
#pragma unroll
      for (int t = MT - 1; t >= 0; --t) {
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
    __builtin_amdgcn_wave_barrier();  // the rows and the data are in LDS
    refresh();

    // ---- the sweep ----
    int frontier = 0, next = 0;  // positions < frontier are committed; positions < next have been handed out
    // per-lane walk state
    bool active = false;
    int pidx = 0, g = 0, g_first = 0;
    double tn = 0.0, ts = 0.0, S = 0.0, B0 = 0.0, thr = 0.0, lu = 0.0;
    uint32_t cfn_hi = 0, cfs_hi = 0;
    int ki = 0, n_new = 0, next_i = ABD_G2_NONE;  // the lane's own list of new kept infections: next entry, count, its position
    int kv = 0, next_v = ABD_G2_NONE;             // the next vaccination at or after the walk's gap
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
            // how the flip of raw bit d changes i0 (abd.py:643-647 / 818 + 771): one position leaves (a_rm), one enters (b_add)
            const bool was_one = ((row_r[d >> 6] >> (d & 63)) & 1ull) != 0;
            const double delta0 = was_one ? -theta0 : theta0;  // Bernoulli(i_raw | p) on the RAW matrix (abd.py:427)
            int a_rm = ABD_G2_NONE, b_add = ABD_G2_NONE;
            if (nch <= 1) {
              const bool pcr_bit = a.pw != nullptr && ((row_p[d >> 6] >> (d & 63)) & 1ull) != 0;
              if (!pcr_bit) {  // where(i_raw + pcrpos > 0, 1, 0): a PCR+ gap is an infection either way
                if (was_one) a_rm = d;
                else b_add = d;
              }
            } else {
              const int cc = (d >= chunk_lo1 ? 1 : 0) + (d >= chunk_lo2 ? 1 : 0);
              const int f1 = cc == 0 ? f1_0 : (cc == 1 ? f1_1 : f1_2), f2 = cc == 0 ? f2_0 : (cc == 1 ? f2_1 : f2_2);
              if (!((chunk_pcr >> cc) & 1)) {  // (a chunk with a PCR+ is that PCR+ column whatever i_raw holds: abd.py:771)
                if (was_one) {
                  if (d == f1) {  // the chunk's first raw infection goes: the second one (if any) takes its place (abd.py:818)
                    a_rm = d;
                    b_add = f2;
                  }
                } else if (d < f1) {  // a raw infection in front of the chunk's first one replaces it
                  b_add = d;
                  a_rm = f1;
                }
              }
            }
            const int p0 = min(a_rm, b_add);
            int gf = ABD_G2_NONE;
            bool overflow = false;
            if (p0 < ABD_G2_NONE) {
              // the three-gap pass (abd.py:560-601) restarted at p0: the kept infections before p0 stay (the pass is causal)
              int block_until = 0, kc = 0;  // kc: the current kept infections ipos[kc ..) lie at or after p0
              for (int k = 0; k < n_i; ++k) {
                const int pos = ipos[k];
                if (pos < p0) {
                  block_until = pos + 4;
                  kc = k + 1;
                }
              }
              // ... goes on over the new i0 from p0 on, compared entry by entry with the current kept infections: the first
              // difference is the first gap whose constrained infection changes
              n_new = 0;
              ki = -1;  // index of the first new entry that differs from the current list
              auto consider = [&](int gc) {
                if (gc >= block_until) {
                  if (n_new < ABD_G2_KCAP) inl[n_new * 64] = (uint16_t)gc;
                  if (ki < 0) {
                    const int cur = kc < n_i ? (int)ipos[kc] : ABD_G2_NONE;
                    if (cur != gc) {
                      gf = min(cur, gc);
                      ki = n_new;
                    } else {
                      ++kc;
                    }
                  }
                  ++n_new;
                  block_until = gc + 4;
                }
              };
              bool b_pending = b_add < ABD_G2_NONE;
              for (int k = 0; k < n_i0; ++k) {
                const int pos = i0pos[k];
                if (pos >= p0 && pos != a_rm) {
                  if (b_pending && b_add < pos) {
                    consider(b_add);
                    b_pending = false;
                  }
                  consider(pos);
                }
              }
              if (b_pending) consider(b_add);
              if (ki < 0 && kc < n_i) {  // the new list ended first: the next current infection is the first one that goes
                gf = ipos[kc];
                ki = n_new;
              }
              overflow = n_new > ABD_G2_KCAP;
            }
            if (gf >= ABD_G2_NONE) {
              // the constrained infections do not change: the prior term decides
              result[pidx] = (delta0 > 0.0 || delta0 > lu) ? ABD_G2_ACCEPT : ABD_G2_REJECT;
            } else if (overflow) {
              result[pidx] = ABD_G2_COMPLEX;  // more new infections than a lane holds: the whole wave evaluates it at the frontier
            } else {
              B0 = delta0 + suf[gf];  // everything the current state holds from gf on is given up
              thr = lu - 1e-9 * (fabs(B0) + 1.0);
              // the two responses at gap gf - 1 of the CURRENT state (the states agree below gf): the dense design
              // (abd.py:258-274) summed over the current exposures before gf, in the order of the packed rows' words
              {
                const double2_t* tsb = wj ? tabs + tstride : tab_ones;
                tn = ts = 0.0;
                for (int k = 0; k < n_e; ++k) {
                  const int e = epos[k];
                  const int pos = e & 0x7FFF;
                  const int idx = pos < gf ? gf - pos : 0;  // 0 = "at or after gf": contributes nothing
                  if (!(e & 0x8000)) tn += tabs[idx].x;
                  ts += tsb[idx].x;
                }
              }
              cfn_hi = firstI < gf ? 0x3FF00000u : 0u;
              cfs_hi = min(firstI, firstV) < gf ? 0x3FF00000u : 0u;
              next_i = ki < n_new ? (int)inl[ki * 64] : ABD_G2_NONE;
              kv = 0;
              for (int k = 0; k < n_v; ++k) kv += (int)vpos[k] < gf ? 1 : 0;
              next_v = kv < n_v ? (int)vpos[kv] : ABD_G2_NONE;
              S = 0.0;
              g = g_first = gf;
              active = true;
            }
          }
        }
        next = min(n_prop, next + n_idle);
      }

      // ---- 2. up to ABD_G2_STEPS gaps for every walking lane (the scheduler's own work per iteration -- ballots, the commit
      // scan, the tail test -- is as much as a gap's; a lane whose walk ends sits the remaining gaps of the iteration out.
      // Measured, config 3, 4 chains, converged / random state: 1 gap per iteration 0.87 / 2.08 ms, 2: 0.80 / 1.82, 4: 0.75 /
      // 1.64, 6: 0.74 / 1.57, 8: 0.75 / 1.54) ----
      bool finished = false;
      for (int rep = 0; rep < ABD_G2_STEPS && __builtin_amdgcn_ballot_w64(active) != 0; ++rep)
      if (active) {
        const bool hit_i = g == next_i, hit_v = g == next_v;
        const uint32_t ei_hi = hit_i ? 0x3FF00000u : 0u;
        const uint32_t ev_hi = hit_v ? 0x3FF00000u : 0u;
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
        if (hit_i) {
          ++ki;
          next_i = ki < n_new ? (int)inl[ki * 64] : ABD_G2_NONE;
        }
        if (hit_v) {
          ++kv;
          next_v = kv < n_v ? (int)vpos[kv] : ABD_G2_NONE;
        }
        ++g;
        // every remaining term is <= 0: delta <= B0 + S; the factor keeps the test on the safe side of rounding
        const bool dead = fma(S, 1.0 - 1e-9, B0) < thr;
        if (dead || g >= G) {
          const double delta = B0 + S;
          result[pidx] = (!dead && (delta > 0.0 || delta > lu)) ? ABD_G2_ACCEPT : ABD_G2_REJECT;
          active = false;
          finished = true;
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
        const int Lw = __builtin_ctzll(old_walkers);
        if (STATS) ++st_tail;
        const int g_l = __builtin_amdgcn_readlane(g, Lw);
        // the walker's infections from its gap on are entries [ki, n_new) of ITS list, the vaccinations vpos[kv ..)
        const bool ci0 = __builtin_amdgcn_readlane((int)cfn_hi, Lw) != 0, civ0 = __builtin_amdgcn_readlane((int)cfs_hi, Lw) != 0;
        double term[MT];
        g2_eval_rounds<R, MT>(G, p, lane, inl - lane0 + Lw, 64, __builtin_amdgcn_readlane(ki, Lw), __builtin_amdgcn_readlane(n_new, Lw), vpos,
                              __builtin_amdgcn_readlane(kv, Lw), n_v, tabs, wj ? tabs + tstride : tab_ones, pwn, wj ? pws_w : 1.0, dataN,
                              dataS, tab_e2, term, g_l, readlane_f64(tn, Lw), readlane_f64(ts, Lw), ci0, civ0);
        double tsum = 0.0;
#pragma unroll
        for (int t = 0; t < MT; ++t) tsum += term[t];
        const double rest = wave_sum_uniform(tsum);
        if (lane == Lw) {
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
          // the whole wave evaluates the proposed state: the waning flip (rho_j changes at every gap), or an i_raw flip with
          // more new infections than a lane holds
          const bool wn = d == G ? !wj : wj;
          double delta_prior;
          const uint16_t* il = ipos;  // the proposed state's infections: the current ones (waning flip) ...
          int nil = n_i;
          if (d == G) {
            delta_prior = wn ? theta7 : -theta7;  // Bernoulli(waner | p_waner) abd.py:373
          } else {
            // ... or those of the flipped raw row, constrained from scratch (abd.py:640-667), as a list
            uint64_t Rn[MT], P[MT], I0n[MT], In[MT], none[MT];
            g2_load_row<MT>(row_r, Rn);
            g2_load_row<MT>(row_p, P);
            bool was_one = false;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
              none[t] = 0;
              if (t == (d >> 6)) {
                was_one = ((Rn[t] >> (d & 63)) & 1ull) != 0;
                Rn[t] ^= 1ull << (d & 63);
              }
            }
            constrain_i0<MT>(Rn, P, nch, cmk, I0n);
            three_gaps_from<MT>(I0n, none, 0, In);
            delta_prior = was_one ? -theta0 : theta0;
            nil = 0;
#pragma unroll
            for (int t = 0; t < MT; ++t) nil = g2_append_positions(In[t], t, lane, tpos, nil, 0);
            il = tpos;
            __builtin_amdgcn_wave_barrier();
          }
          double term[MT];
          g2_eval_rounds<R, MT>(G, p, lane, il, 1, 0, nil, vpos, 0, n_v, tabs, wn ? tabs + tstride : tab_ones, pwn, wn ? pws_w : 1.0, dataN, dataS,
                                tab_e2, term);
          double tsum = 0.0;
#pragma unroll
          for (int t = 0; t < MT; ++t) tsum += term[t];
          const double delta = delta_prior + (wave_sum_uniform(tsum) - total_cur);
          const double log_u = readfirstlane_f64(log_uniform_u32(accw[d], tab_e2));
          accepted = delta > 0.0 || delta > log_u;
          if (accepted && d == G) wj = wn;
        }
        if (accepted && d < G && lane == 0) row_r[d >> 6] ^= 1ull << (d & 63);
        frontier += 1;
        if (!accepted) continue;
        // the state has changed: everything evaluated beyond this proposal is void
        ++n_acc;
        for (int e = frontier + lane; e < next; e += 64) result[e] = ABD_G2_PENDING;
        next = frontier;
        active = false;
        __builtin_amdgcn_wave_barrier();
        refresh();
        break;
      }
    }
    n_prop_total += (unsigned int)frontier;

    // ---- write the individual's state back: raw bits, waning flag, and what the slot keeps beside them (the
    // constrained words the evaluation kernels read, the changes of sum(i_raw) and sum(ab_s_waner)) ----
    int pc1 = wj ? (1 << 16) : 0;
    {
      uint64_t Rw[MT];
      g2_load_row<MT>(row_r, Rw);
#pragma unroll
      for (int t = 0; t < MT; ++t) pc1 += __builtin_popcountll(Rw[t]);
    }
    if (lane < nt) {
      rw[(int64_t)lane * N + j] = row_r[lane];
      iw[(int64_t)lane * N + j] = row_i[lane];
    }
    if (lane == 0) waner[j] = wj ? 1 : 0;
    d_n1 += (pc1 & 0xFFFF) - (pc0 & 0xFFFF);
    d_m1 += (pc1 >> 16) - (pc0 >> 16);
    __builtin_amdgcn_wave_barrier();
  }
  if (lane0 == 0 && (d_n1 | d_m1)) {
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(const_cast<long long*>(cp.cnt));
    atomicAdd(cnt + 0, (unsigned long long)(long long)d_n1);  // two's complement: a negative change wraps to the right sum
    atomicAdd(cnt + 1, (unsigned long long)(long long)d_m1);
  }
  if (lane0 == 0 && (n_acc | n_prop_total)) {
    atomicAdd(ga.counts + 2 * c + 0, (unsigned long long)n_acc);
    atomicAdd(ga.counts + 2 * c + 1, (unsigned long long)n_prop_total);
  }
  if (STATS && lane0 == 0 && ga.stats) {
    atomicAdd(ga.stats + 0, st_inds);
    atomicAdd(ga.stats + 1, st_iter);
    atomicAdd(ga.stats + 2, st_refill);
    atomicAdd(ga.stats + 3, st_steps);
    atomicAdd(ga.stats + 4, st_lane_steps);
    atomicAdd(ga.stats + 5, st_tail);
    atomicAdd(ga.stats + 6, st_commit);
    atomicAdd(ga.stats + 7, (unsigned long long)n_acc);
  }
}
