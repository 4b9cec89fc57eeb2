// abd_dense.hpp -- dense-panel evaluation kernel (exactly one S and one N reading in every (gap, individual) cell).
//
// lane = individual.  The (lane group, gap) plane -- n_lg groups of 64 individuals x G gap rows -- is cut
// into equal contiguous ranges, one per wave "slot", so every slot walks the same number of gap rows
// (+-1) whatever N and G are, and the grid is an exact multiple of the CU count.  A range is walked as
// one or more pieces (it may end one lane group and begin the next).  Inside a piece the responses advance by the
// recurrence the reference tests as equivalent (abd.py:277-293); a piece that does not start at gap 0 rebuilds its
// start state per lane with the reference's dense design (abd.py:258-274) summed over the set bits before the piece,
// rho^k read from the chain's LDS table.
//
// What is read: the two OD panels (gap rows, 1 KiB per wave and antigen), the chain's CONSTRAINED infection words
// `iw` -- the Deterministic "i" (abd.py:640-667), kept per chain slot and refreshed by whoever rewrites the slot's
// discrete state (abd_small.hpp: abd_constrain_kernel; the sweep kernels) -- and the vaccination words, 32 gaps at a
// time; nothing of the integer pre-pass runs here.  sum(i_raw) and sum(ab_s_waner) come from the slot's counters.
//
// The 4 waves of a workgroup are CB chains x (4 / CB) neighbouring ranges: with CB = 4 the panel rows
// they share are fetched from HBM once and served from L1/L2 to the other three.
//
// Latency of one launch matters as much as its throughput (a NUTS chain waits for every evaluation, abd.py:922), so
// the set-up issues every memory access it needs -- range, 2^(j/1024) table, the piece's words, its first gap rows --
// before it builds the power tables, and the gap rows are prefetched two gaps ahead across the whole piece.
//
// The kernel is bound by vector-instruction issue (fp64 instructions issue at half the rate of 32-bit ones),
// so the gap loop is written for instruction count:
//   * the chain constants are pre-scaled by c = 1024 b log2(e): t = c (a - x) is fma(-c, x, c a) with
//     c a = fma(c temp, U, fma(cf, c perm, c init)); the sum for d/db is sum h t (the host divides by c)
//   * indicator bits -> 0.0 / 1.0 doubles by v_bfe_i32 (scalar bit index) + v_and with the high word of 1.0;
//     the "exposed so far" flags of perm_response (abd.py:306) are integer ORs of those high words
//   * e^u = 2^(t/1024), t = 1024 e + j + f: T[j] = 2^(j/1024) from a 1024-entry LDS table, a cubic in f,
//     the exponent e added into T's high word; 1 + 2^t is ONE fma (tools/exp2_table.py: 3.5e-16)
//   * one v_rcp_f64 per cell for both antigens (1/A = B/(AB))
//   * gap rows are addressed as scalar row offset + a constant per-lane offset (no vector address arithmetic)
#pragma once

#include <cstddef>

#include "abd_device.hpp"
#include "abd_terms.hpp"

// One {od, log_dilution} pair of a gap row: buffer load with a scalar row offset and a constant per-lane offset.
typedef uint32_t abd_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t abd_u32x2 __attribute__((ext_vector_type(2)));
template <typename R>
__device__ __forceinline__ YX<R> load_yx(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff);
template <>
__device__ __forceinline__ YX<double> load_yx<double>(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff) {
  const abd_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  YX<double> r;
  r.y = __hiloint2double((int)v.y, (int)v.x);
  r.x = __hiloint2double((int)v.w, (int)v.z);
  return r;
}
template <>
__device__ __forceinline__ YX<float> load_yx<float>(const __amdgpu_buffer_rsrc_t& rs, uint32_t voff, uint32_t soff) {
  const abd_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
  YX<float> r;
  r.y = __uint_as_float(v.x);
  r.x = __uint_as_float(v.y);
  return r;
}

// ---- where a gap row's {od, log_dilution} comes from ----
// XC = false: the pair panel, [G][N] of {od, log_dilution} in the storage type: 2 R bytes per cell and antigen.
// XC = true:  the split panels of a launch that evaluates ONE chain and therefore reads every byte for that chain alone (a
//             NUTS chain's leapfrogs, abd.py:922; BASELINE config 5): od [lane group][G][64] in the storage type (the rows a
//             wave walks are one contiguous stream) + a one-byte code per
//             cell into the antigen's dictionary of distinct log dilutions (assays use a handful of dilutions; abd_create
//             builds the code panel when an antigen has <= 256 distinct values) -- R + 1 bytes per cell and antigen, lossless;
//             the dictionary (<= 2 KB) sits in LDS.  Four independent chains at config 3 are bound by these bytes.
template <typename R, bool XC>
struct RowData;
template <typename R>
struct RowData<R, false> {
  YX<R> v;
};
template <typename R>
struct RowData<R, true> {
  R y;
  uint32_t code;
};
template <bool XC>
struct RowDesc;  // buffer descriptors of one antigen's panel(s), based at a piece's first row (wave-uniform)
template <>
struct RowDesc<false> {
  __amdgpu_buffer_rsrc_t rs;
};
template <>
struct RowDesc<true> {
  __amdgpu_buffer_rsrc_t rs_y, rs_c;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void* base, int64_t byte_off) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + byte_off, 0, -1, 0x00020000);
}
// (built outside divergent code, or every load through a descriptor becomes a waterfall loop)
template <typename R, bool XC, typename ARGS>
__device__ __forceinline__ RowDesc<XC> row_desc(const ARGS& a, int antigen, int lg, int g0) {
  RowDesc<XC> d;
  if constexpr (XC) {  // lane-group-major: [lane group][gap][64]
    const int64_t cell0 = ((int64_t)lg * a.G + g0) * 64;
    d.rs_y = rsrc_at(antigen ? a.od_s : a.od_n, cell0 * (int64_t)sizeof(R));
    d.rs_c = rsrc_at(antigen ? a.xc_s : a.xc_n, cell0);
  } else {             // gap-major: [gap][individual]
    const int64_t cell0 = (int64_t)g0 * a.N + (int64_t)lg * 64;
    d.rs = rsrc_at(antigen ? a.yx_s : a.yx_n, cell0 * (int64_t)sizeof(YX<R>));
  }
  return d;
}
// row `row` (relative to the descriptor's base) for this lane: scalar row offset + a constant per-lane offset
template <typename R, bool XC>
__device__ __forceinline__ RowData<R, XC> row_load(const RowDesc<XC>& d, int lane, int row, int N) {
  RowData<R, XC> r;
  if constexpr (XC) {
    const uint32_t soff = (uint32_t)row * 64u;
    if constexpr (sizeof(R) == 8) {
      const abd_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(d.rs_y, (uint32_t)lane * 8u, soff * 8u, 0);
      r.y = __hiloint2double((int)v.y, (int)v.x);
    } else {
      r.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d.rs_y, (uint32_t)lane * 4u, soff * 4u, 0));
    }
    r.code = __builtin_amdgcn_raw_buffer_load_b8(d.rs_c, (uint32_t)lane, soff, 0);
  } else {
    r.v = load_yx<R>(d.rs, (uint32_t)lane * (uint32_t)sizeof(YX<R>), (uint32_t)row * (uint32_t)N * (uint32_t)sizeof(YX<R>));
  }
  return r;
}
template <typename R, bool XC>
__device__ __forceinline__ double row_y(const RowData<R, XC>& r) {
  if constexpr (XC) return (double)r.y;
  else return (double)r.v.y;
}
template <typename R, bool XC>
__device__ __forceinline__ double row_x(const RowData<R, XC>& r, const double* dict /* LDS */) {
  if constexpr (XC) return dict[r.code];
  else return (double)r.v.x;
}

// 32 gaps (word q: gaps 32 q .. 32 q + 31) of one individual's packed row.  The panels are [64-gap word][individual]
// arrays of 64-bit words; base = the panel seen as 32-bit halves (wave-uniform), j2 = 2 x the lane's individual: the
// address is a scalar base + a 32-bit lane offset, no 64-bit vector arithmetic.
__device__ __forceinline__ uint32_t word32(const uint32_t* base, uint32_t j2, int q, int N) {
  const uint32_t* row = base + ((int64_t)(q >> 1) * 2 * N + (q & 1));
  return row[j2];
}

// 1 + 2^(t / 1024) for both antigens of a cell, as one_plus_exp2_tab (abd_device.hpp) twice -- written stage by stage for
// the two independent chains, so that a wave that has its SIMD to itself (a synchronous call, a sampler unit: one wave per
// SIMD) overlaps their latencies (fp64 results take ~2 issue slots to come back, the table read ~16) instead of walking one
// chain after the other.  The polynomial's second coefficient is held in a vector register by the caller (c2v): a VOP3 fma
// takes one scalar operand, and left alone the compiler re-materialises that constant with a v_mov_b64 per evaluation.
#define ABD_EXP2_C2 0x1.ebfbe033445b4p-23
__device__ __forceinline__ void one_plus_exp2_pair(double t_n, double t_s, const double* tab /* LDS */, double c2v, double& A,
                                                   double& B) {
  const double kf_n = __builtin_rint(t_n), kf_s = __builtin_rint(t_s);
  const double f_n = t_n - kf_n, f_s = t_s - kf_s;
  int k_n, k_s;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(k_n) : "v"(kf_n));  // saturating; a C++ cast of an out-of-range double is undefined
  asm("v_cvt_i32_f64 %0, %1" : "=v"(k_s) : "v"(kf_s));
  const double T_n = tab[k_n & (ABD_EXP2_TAB - 1)], T_s = tab[k_s & (ABD_EXP2_TAB - 1)];
  double p_n, p_s;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p_n) : "s"(0x1.c6b08d910ecbdp-35), "v"(f_n), "v"(c2v));  // tools/exp2_table.py 1024 3
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p_s) : "s"(0x1.c6b08d910ecbdp-35), "v"(f_s), "v"(c2v));
  const int e_n = min(max(k_n >> 10, -1022), 510), e_s = min(max(k_s >> 10, -1022), 510);  // v_med3_i32
  p_n = fma(p_n, f_n, 0x1.62e42fefa39efp-11);
  p_s = fma(p_s, f_s, 0x1.62e42fefa39efp-11);
  p_n = fma(p_n, f_n, 1.0);
  p_s = fma(p_s, f_s, 1.0);
  const double Ts_n = __hiloint2double(__double2hiint(T_n) + (e_n << 20), __double2loint(T_n));  // T 2^e: v_lshl_add_u32
  const double Ts_s = __hiloint2double(__double2hiint(T_s) + (e_s << 20), __double2loint(T_s));
  A = fma(Ts_n, p_n, 1.0);
  B = fma(Ts_s, p_s, 1.0);
}

// Both antigens of one cell at once: the two reciprocals 1/(1+e_n), 1/(1+e_s) come from ONE v_rcp_f64 (quarter
// rate) of the product -- 1/A = B/(AB), 1/B = A/(AB).  t_n = 1024 log2(e) b_n (a_n - x_n), t_s likewise.
template <bool GRAD>
__device__ __forceinline__ void obs_pair_scaled(double t_n, double yn, double d_n, double t_s, double ys, double d_s,
                                                const double* tab, double c2v, double (&acc)[16], double& h_n, double& h_s) {
  double A, B;
  one_plus_exp2_pair(t_n, t_s, tab, c2v, A, B);
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
    acc[A_N_HX] = fma(h_n, t_n, acc[A_N_HX]);  // c_n sum h (a - x): the host divides by c_n (abd_context.hip: assemble)
    acc[A_S_HX] = fma(h_s, t_s, acc[A_S_HX]);
  }
}

#ifdef ABD_STAMPS
// diagnostic build only (tools/probe_stamps.py): wave 0 of a few workgroups records s_memrealtime (100 MHz) at phase
// boundaries into EvalArgs::stamps[blockIdx.x][16]; never compiled into the product
#define ABD_STAMP(k)                                                                                  \
  do {                                                                                                \
    if (a.stamps && wave == 0 && lane == 0 && blockIdx.y == 0)                                        \
      a.stamps[(int64_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime();                   \
  } while (0)
#else
#define ABD_STAMP(k)
#endif

// The chain's constants as the gap loop wants them, pre-scaled by c = 1024 log2(e) b (wave-uniform; a VOP3 fma reads at
// most one scalar operand, so the addends that meet another chain constant live in VGPRs, or every use costs a v_mov_b64).
struct DenseChain {
  double rho_n, rho_s;
  double ct_n, cp_n, ci_n, mc_n;  // c_n temp_n, c_n perm_n, c_n init_n (VGPR), -c_n
  double c_s, cp_s, ci_s, mc_s;   // c_s (unit boosts: abd.py:272), c_s perm_s, c_s init_s (VGPR), -c_s
  double d_n, d_s;
};
__device__ __forceinline__ DenseChain dense_chain(const ChainPar& p) {
  // the products are computed by the vector unit (there is no scalar fp64 multiply) and moved back into scalar registers:
  // left in VGPRs, eight wave-uniform doubles cost the loop 16 registers it does not have
  DenseChain k;
  const double c_n = p.b_n * (1.4426950408889634074 * ABD_EXP2_TAB), c_s = p.b_s * (1.4426950408889634074 * ABD_EXP2_TAB);
  k.rho_n = p.rho_n;
  k.rho_s = p.rho_s;
  k.ct_n = readfirstlane_f64(c_n * p.temp_n);
  k.cp_n = readfirstlane_f64(c_n * p.perm_n);
  k.ci_n = to_vgpr(c_n * p.init_n);
  k.mc_n = readfirstlane_f64(-c_n);
  k.c_s = readfirstlane_f64(c_s);
  k.cp_s = readfirstlane_f64(c_s * p.perm_s);
  k.ci_s = to_vgpr(c_s * p.init_s);
  k.mc_s = readfirstlane_f64(-c_s);
  k.d_n = p.d_n;
  k.d_s = p.d_s;
  return k;
}

// What a piece (lane group lg, gaps [g0, g1)) needs from memory before its walk can start, requested as early as possible:
// the words before g0 (start state), the words of its first gaps, its first gap rows.
#define ABD_SW 8  // 32-gap words per group of the start-state pass (covers g0 <= 256 in one group)
template <typename R, bool XC>
struct PieceLoads {
  uint32_t seg_i, seg_v, nxt_i, nxt_v;  // words g0 >> 5 and (g0 >> 5) + 1
  RowData<R, XC> en, es, on, os;    // gap rows of the first even and the first odd gap of the piece (N, S antigen)
  int wj;                           // ab_s_waner of the lane's individual
};

template <typename R, bool XC, typename ARGS>
__device__ __forceinline__ void piece_issue_loads(const ARGS& a, const RowDesc<XC>& rs_n, const RowDesc<XC>& rs_s,
                                                  const uint32_t* ibase, const uint32_t* vbase, const int8_t* waner, int lane, int j,
                                                  int g0, int g1, PieceLoads<R, XC>& pl) {
  const int N = a.N;
  const int q = g0 >> 5, q_end = (g1 - 1) >> 5;
  // the rows first: they come from the farthest away (G rows of a panel fit a 32-bit offset: abd_create checks)
  const int last = g1 - 1 - g0;
  const int re = min(g0 & 1, last), ro = min((g0 & 1) ^ 1, last);  // row (relative to g0) of the first even / odd gap
  if (g0 & 1) {
    pl.on = row_load<R, XC>(rs_n, lane, ro, N);
    pl.os = row_load<R, XC>(rs_s, lane, ro, N);
    pl.en = row_load<R, XC>(rs_n, lane, re, N);
    pl.es = row_load<R, XC>(rs_s, lane, re, N);
  } else {
    pl.en = row_load<R, XC>(rs_n, lane, re, N);
    pl.es = row_load<R, XC>(rs_s, lane, re, N);
    pl.on = row_load<R, XC>(rs_n, lane, ro, N);
    pl.os = row_load<R, XC>(rs_s, lane, ro, N);
  }
  const uint32_t j2 = 2u * (uint32_t)j;
  pl.seg_i = word32(ibase, j2, q, N);
  pl.seg_v = word32(vbase, j2, q, N);
  pl.nxt_i = pl.nxt_v = 0;
  if (q < q_end) {
    pl.nxt_i = word32(ibase, j2, q + 1, N);
    pl.nxt_v = word32(vbase, j2, q + 1, N);
  }
  pl.wj = waner[j];
}

// words 0 .. ABD_SW - 1 of the constrained infections / the vaccinations, as far as they lie before g0 > 0 (start state)
__device__ __forceinline__ void state_issue_loads(const uint32_t* ibase, const uint32_t* vbase, uint32_t j2, int N, int g0,
                                                  uint32_t (&wi)[ABD_SW], uint32_t (&wv)[ABD_SW]) {
  const int qlast = (g0 - 1) >> 5;
#pragma unroll
  for (int u = 0; u < ABD_SW; ++u) {
    wi[u] = wv[u] = 0;
    if (u <= qlast) {
      wi[u] = word32(ibase, j2, u, N);
      wv[u] = word32(vbase, j2, u, N);
    }
  }
}

// State of one lane's individual at the end of gap g0 - 1 (g0 > 0): the dense design (abd.py:258-274) summed over the
// exposures before the piece, rho^k from the chain's LDS tables: entry e of a table = {rho^(e-1), (e-1) rho^(e-2)}, entry 0
// = {0, 0}.  Individuals whose S response does not wane (rho_j = 1, abd.py:374) count their exposures instead.
__device__ __forceinline__ void dense_start_state(uint32_t (&wi)[ABD_SW], uint32_t (&wv)[ABD_SW], const uint32_t* ibase,
                                                  const uint32_t* vbase, uint32_t j2, int N, int g0, const double2_t* tab_n,
                                                  const double2_t* tab_s, bool wj, double& tn, double& dn, double& ts, double& ds,
                                                  uint32_t& cfn_hi, uint32_t& cfs_hi) {
  const int qlast = (g0 - 1) >> 5;
  uint32_t any_i = 0, any_v = 0;
  int cnt = 0;
  for (int q0 = 0;; q0 += ABD_SW) {
#pragma unroll
    for (int u = 0; u < ABD_SW; ++u) {
      const int q = q0 + u;
      if (q <= qlast) {                 // wave-uniform
        const int rel = g0 - q * 32;    // bits < rel of this word are before the piece (rel >= 1)
        uint32_t mi = wi[u], mv = wv[u];
        if (rel < 32) {
          const uint32_t below = (1u << rel) - 1u;
          mi &= below;
          mv &= below;
        }
        any_i |= mi;
        any_v |= mv;
        cnt += __builtin_popcount(mi) + __builtin_popcount(mv);
        // one infection and one vaccination per turn (table entry 0 = {0, 0} stands in where a lane has none left): the
        // turns of a word are as many as the larger of the two counts, and each turn waits for LDS once
        if (__builtin_amdgcn_ballot_w64((mi | mv) != 0) != 0) {
          while (mi | mv) {  // per-lane trip count
            const int idx_i = mi ? rel - __builtin_ctz(mi) : 0;  // = k + 1 with k = (g0 - 1) - gap of the bit
            const int idx_v = mv ? rel - __builtin_ctz(mv) : 0;
            mi &= mi - 1;  // (0 stays 0)
            mv &= mv - 1;
            const double2_t pn = tab_n[idx_i];
            const double2_t ps = tab_s[idx_i];
            const double2_t pv = tab_s[idx_v];
            tn += pn.x;
            dn += pn.y;
            ts += ps.x;
            ds += ps.y;
            ts += pv.x;
            ds += pv.y;
          }
        }
      }
    }
    if (q0 + ABD_SW > qlast) break;
    // more than ABD_SW words before the piece (g0 > 256): the next group
#pragma unroll
    for (int u = 0; u < ABD_SW; ++u) {
      const int q = q0 + ABD_SW + u;
      wi[u] = wv[u] = 0;
      if (q <= qlast) {
        wi[u] = word32(ibase, j2, q, N);
        wv[u] = word32(vbase, j2, q, N);
      }
    }
  }
  if (!wj) {
    ts = (double)cnt;  // unit boosts that never wane: an infection and a dose in the same gap both count (Q5)
    ds = 0.0;
  }
  cfn_hi = any_i ? 0x3FF00000u : 0u;
  cfs_hi = (any_i | any_v) ? 0x3FF00000u : 0u;
}

// Walk gaps [g0, g1) of lane group lg: recurrence form (abd.py:288) + likelihood terms into acc.
template <typename R, bool GRAD, bool XC, typename ARGS>
__device__ __forceinline__ void dense_walk(const ARGS& a, const DenseChain& k, const RowDesc<XC>& rs_n,
                                           const RowDesc<XC>& rs_s, const double* dict_n, const double* dict_s,
                                           const uint32_t* ibase, const uint32_t* vbase,
                                           uint32_t j2, PieceLoads<R, XC>& pl, int lane, int g0, int g1, bool wj, double tn, double dn, double ts,
                                           double ds, uint32_t cfn_hi, uint32_t cfs_hi, const double* tab_e2, double c2v,
                                           double (&acc)[16]) {
  const int N = a.N;
  const double rho_n = k.rho_n, ct_n = k.ct_n, cp_n = k.cp_n, ci_n = k.ci_n, mc_n = k.mc_n;
  const double c_s = k.c_s, cp_s = k.cp_s, ci_s = k.ci_s, mc_s = k.mc_s, d_n = k.d_n, d_s = k.d_s;
  const double rho_j = wj ? k.rho_s : 1.0;  // abd.py:374
  double hd_s = 0.0;
  const uint32_t z_ei = zero_vgpr(), z_ev = zero_vgpr(), z_cn = zero_vgpr(), z_cs = zero_vgpr();
  const int last = g1 - 1 - g0;
  // gap rows through buffer loads: descriptor base = the piece's first row of this lane group, scalar offset = row
  // within the piece, vector offset = the lane's constant -- no vector address arithmetic at all
  auto ldrow = [&](const RowDesc<XC>& rs, int g) { return row_load<R, XC>(rs, lane, min(g - g0, last), N); };
  int q = g0 >> 5;
  const int q_end = (g1 - 1) >> 5;
  uint32_t seg_i = pl.seg_i, seg_v = pl.seg_v;
  // the next 32 gaps' indicator words arrive while this word's gaps are walked
  auto next_word = [&]() {
    seg_i = pl.nxt_i;
    seg_v = pl.nxt_v;
    ++q;
    if (q < q_end) {
      pl.nxt_i = word32(ibase, j2, q + 1, N);
      pl.nxt_v = word32(vbase, j2, q + 1, N);
    }
  };

  // one gap: 0/1 indicators enter as doubles, so the perm switch (abd.py:306) is an fma, not a select
  auto step = [&](int g, const RowData<R, XC>& on, const RowData<R, XC>& os) {
    const uint32_t bit = (uint32_t)g & 31u;
    const uint32_t ei_hi = (uint32_t)__builtin_amdgcn_sbfe((int)seg_i, bit, 1u) & 0x3FF00000u;
    const uint32_t ev_hi = (uint32_t)__builtin_amdgcn_sbfe((int)seg_v, bit, 1u) & 0x3FF00000u;
    const double e_i = hi_to_double(ei_hi, z_ei), e_v = hi_to_double(ev_hi, z_ev);
    cfn_hi |= ei_hi;
    cfs_hi |= ei_hi | ev_hi;
    dn = fma_s(rho_n, dn, tn);
    tn = fma_s(rho_n, tn, e_i);
    ds = fma_v(rho_j, ds, ts);
    ts = fma_v(rho_j, ts, e_i + e_v);  // unit boosts: temp unused (abd.py:272)
    const double cf_n = hi_to_double(cfn_hi, z_cn), cf_s = hi_to_double(cfs_hi, z_cs);
    // c mu_n = c (perm + temp + init)   abd.py:341 ; c mu_s = c (perm + tinf + tvac + init)   abd.py:389-391
    const double t_n = fma(mc_n, row_x<R, XC>(on, dict_n), fma(ct_n, tn, fma(cf_n, cp_n, ci_n)));
    const double t_s = fma(mc_s, row_x<R, XC>(os, dict_s), fma(c_s, ts, fma(cf_s, cp_s, ci_s)));
    double h_n = 0.0, h_s = 0.0;
    obs_pair_scaled<GRAD>(t_n, row_y<R, XC>(on), d_n, t_s, row_y<R, XC>(os), d_s, tab_e2, c2v, acc, h_n, h_s);
    if (GRAD) {
      acc[A_N_HC] = fma(h_n, cf_n, acc[A_N_HC]);
      acc[A_N_HU] = fma(h_n, tn, acc[A_N_HU]);
      acc[A_N_HD] = fma(h_n, dn, acc[A_N_HD]);
      acc[A_S_HC] = fma(h_s, cf_s, acc[A_S_HC]);
      hd_s = fma(h_s, ds, hd_s);
    }
  };

  // two row buffers per antigen, one for the even and one for the odd gaps; each is refilled right after the step that
  // consumed it, for the gap two on -- across word boundaries, to the end of the piece
  RowData<R, XC> en = pl.en, es = pl.es, on = pl.on, os = pl.os;
  int g = g0;
  if (g & 1) {
    step(g, on, os);
    on = ldrow(rs_n, g + 2);
    os = ldrow(rs_s, g + 2);
    ++g;
    if (g < g1 && (g & 31) == 0) next_word();
  }
  // g is even from here on, so a pair of gaps never straddles a 32-gap word: the inner loop walks the pairs of one word
  // with the word's registers loop-invariant (one flat loop with the word change inside cost two register moves per gap)
  while (g + 1 < g1) {
    const int pair_end = min(((g >> 5) + 1) << 5, g1 - 1);  // pairs start below this
    for (; g < pair_end; g += 2) {
      step(g, en, es);
      en = ldrow(rs_n, g + 2);
      es = ldrow(rs_s, g + 2);
      step(g + 1, on, os);
      on = ldrow(rs_n, g + 3);
      os = ldrow(rs_s, g + 3);
    }
    if (g < g1 && (g & 31) == 0) next_word();
  }
  if (g < g1) step(g, en, es);  // (its word is in place: the loop above changed it, or the piece is this one gap)
  acc[A_S_HD] += wj ? hd_s : 0.0;  // d rho_j / d rho_s = waner_j
}

// ---- leapfrog train (abd_types.hpp: TrainArgs): the launch's last workgroup, wave 0, with the 16 sums in sm[0 .. 15] ----
// Lane k < 17 owns value variable k: (1) its share of logp and its gradient entry from the sums and the point's closed-form
// terms (abd_terms.hpp: assemble_lane), logp = the 17 shares added in order; (2) it finishes the leapfrog that led to this
// point and takes the drift of the next one -- the host's arithmetic (abd_nuts.hpp: feed / stage_leapfrog, diagonal
// metric), operation by operation and without contraction, so that the host, which repeats it on the record, arrives at
// the same bits; (3) it transforms the next point and leaves it in slots[next_slot] for the launch queued behind this one;
// (4) the record goes to mapped host memory, its tag last.  sm: >= 64 doubles of LDS.
// the point a train launch evaluates: left in device memory by its predecessor, or staged by the host in the kernel
// arguments (read through the argument segment's address: a lane-indexed access to a by-value struct goes to scratch)
__device__ __forceinline__ const char* train_kernarg() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const char*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  return nullptr;  // (host pass of the compiler: never executed)
#endif
}
__device__ __forceinline__ const TrainPoint* train_point(const TrainArgs& T) {
  return T.use_slot >= 0 ? T.slots + T.use_slot
                         : reinterpret_cast<const TrainPoint*>(train_kernarg() + offsetof(EvalArgs, train) + offsetof(TrainArgs, first));
}

// a successor's first duty (one wave of its first workgroup): the predecessor's record goes to the host -- the
// predecessor left it beside the point (TrainPoint::prev_*) instead of waiting for its own writes to cross PCIe
__device__ __forceinline__ void train_forward_record(const TrainArgs& T, int lane) {
  const TrainPoint* cur = T.slots + T.use_slot;
  if (lane < ABD_NT) {
    T.fwd_rec->g[lane] = cur->prev_g[lane];
    T.fwd_rec->next_theta[lane] = cur->theta[lane];
    T.fwd_rec->next_p_half[lane] = cur->p_half[lane];
    if (lane == 0) T.fwd_rec->lp = cur->prev_lp;
  }
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) __hip_atomic_store(&T.fwd_rec->tag, T.fwd_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void train_epilogue(const EvalArgs& a, double* sm, int lane) {
  const TrainArgs& T = a.train;
  const TrainPoint* cur = train_point(T);
  const double* inv_mass = reinterpret_cast<const double*>(train_kernarg() + offsetof(EvalArgs, train) + offsetof(TrainArgs, inv_mass));
  double* lps = sm + 16;  // [17] the variables' shares of logp
  const int k = lane < ABD_NT ? lane : 0;
  const int q4 = k == 0 ? 0 : k == 3 ? 1 : k == 6 ? 2 : k == 7 ? 3 : -1;
  abdi::Transformed tr;  // (field by field: an indexed write would put the struct in scratch)
  tr.p = cur->tr[0];
  tr.perm_n = cur->tr[1];
  tr.temp_n = cur->tr[2];
  tr.rho_n = cur->tr[3];
  tr.init_n = cur->tr[4];
  tr.perm_s = cur->tr[5];
  tr.rho_s = cur->tr[6];
  tr.q = cur->tr[7];
  tr.tinf = cur->tr[8];
  tr.tvac = cur->tr[9];
  tr.init_s = cur->tr[10];
  tr.b_n = cur->tr[11];
  tr.d_n = cur->tr[12];
  tr.sig_n = cur->tr[13];
  tr.b_s = cur->tr[14];
  tr.d_s = cur->tr[15];
  tr.sig_s = cur->tr[16];
  abdi::ModelSizes m;
  m.G = a.G;
  m.dense = T.dense;
  m.N = (double)a.N;
  m.cells = (double)a.G * (double)a.N;
  m.Kn = (double)a.K_n;
  m.Ks = (double)a.K_s;
  m.prior_const = T.prior_const;
  const double tk = cur->theta[k];
  const double l0 = q4 >= 0 ? cur->L0[q4] : 0.0, l1 = q4 >= 0 ? cur->L1[q4] : 0.0;
  double lp_k, g_k;
  abdi::assemble_lane(k, m, tr, tk, cur->tr[k], l0, l1, sm, lp_k, g_k);
  if (lane < ABD_NT) lps[lane] = lp_k;
  __builtin_amdgcn_wave_barrier();
  double lp = 0.0;
#pragma unroll
  for (int q = 0; q < ABD_NT; ++q) lp += lps[q];  // every lane, same order
  if (lane < ABD_NT) {
    if (T.own_record) {
      T.rec->g[lane] = g_k;  // the record's first half is on its way while the next point is worked out
      if (lane == 0) T.rec->lp = lp;
    }
    const bool finite = __builtin_isfinite(lp);
    const double gd = finite ? g_k : 0.0;                              // feed: cur.g = finite ? g1 : 0
    const double kick = __dmul_rn(__dmul_rn(0.5, T.ve), gd);           // 0.5 * ve * g
    const double p = __dadd_rn(cur->p_half[lane], kick);               // feed: cur.p = p_half + 0.5 ve g
    const double ph = __dadd_rn(p, kick);                              // stage_leapfrog: p_half = cur.p + 0.5 ve cur.g
    const double v = __dmul_rn(inv_mass[lane], ph);                    // velocity, diagonal metric
    const double t2 = __dadd_rn(tk, __dmul_rn(T.ve, v));               // req_q = cur.q + ve v
    if (T.own_record) {
      T.rec->next_theta[lane] = t2;
      T.rec->next_p_half[lane] = ph;
    }
    double tr2, n0, n1;
    abdi::transform_lane(lane, t2, tr2, n0, n1);
    TrainPoint* nx = T.slots + T.next_slot;
    nx->theta[lane] = t2;
    nx->p_half[lane] = ph;
    nx->tr[lane] = tr2;
    if (q4 >= 0) {
      nx->L0[q4] = n0;
      nx->L1[q4] = n1;
    }
    nx->prev_g[lane] = g_k;  // the successor passes this launch's result on to the host
    if (lane == 0) nx->prev_lp = lp;
  }
  if (!T.own_record) return;
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) __hip_atomic_store(&T.rec->tag, T.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// dynamic LDS of the kernel: [CB][2][G+1] power tables, block reduction, 2^(j/1024) table (and at least the scratch of the
// fused fixed-order sum; a train launch's last workgroup also keeps the sums of its chains there: abd_train.hpp)
#define ABD_TRAIN_SM 64  // doubles of LDS per chain of a train launch's last workgroup: 16 sums, 17 shares of logp
__host__ __device__ inline size_t abd_dense_lds(int G, int cb, bool xc = false, bool train = false) {
  const size_t need = (size_t)cb * 2 * (size_t)(G + 1) * sizeof(double2_t) + (size_t)ABD_WAVES_PER_BLOCK * ABD_NOUT * sizeof(double) +
                      (size_t)ABD_EXP2_TAB * sizeof(double) + (xc ? (size_t)2 * ABD_XDICT * sizeof(double) : 0);
  const size_t fin = (size_t)ABD_FIN_PARTS * ABD_NOUT * sizeof(double) + (train ? (size_t)ABD_TRAIN_CB * ABD_TRAIN_SM * sizeof(double) : 0);
  return need > fin ? need : fin;
}

template <typename ARGS>
struct is_train_args {
  static constexpr bool value = false;
};
template <>
struct is_train_args<DenseTrainArgs> {
  static constexpr bool value = true;
};

// (abd_train.hpp)
__device__ __forceinline__ void train_service(const DenseTrainArgs& a, int wave, int lane);
__device__ __forceinline__ void train_step(const DenseTrainArgs& a, const TrainChainArgs& tc, double* sm, int lane);

// Count-in in two levels and the fixed-order sums of a launch's partial rows (see dense_body).  rows: [CB][nblk][ABD_NOUT] partial
// rows of this workgroup's CB chains, this workgroup's already stored write-through by wave 0 in front of a workgroup barrier;
// shard_rows: [CB][ABD_TRAIN_SHARDS][ABD_NOUT]; cnt: [0] the top counter, [(1 + s) * ABD_TRAIN_CNT_STRIDE] shard s (all zero between
// launches); mask: bit cc = chain cc has rows to sum.  Returns true in ONE workgroup of the launch -- the one whose count came
// last at the top -- with chain cc's 16 sums in sm_chain[cc * ABD_TRAIN_SM ..]; every other workgroup gets false and is done.
// flag: one int of LDS.  Fixed orders (a shard's rows in range order, the shards in order): the sums depend on nblk only.
template <int CB>
__device__ __forceinline__ bool two_level_sums(const double* rows, double* shard_rows, unsigned int* cnt, int nblk, int blk, unsigned int mask,
                                               int* flag, double* sm_chain, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int shard = blk % ABD_TRAIN_SHARDS;
  const int n_in_shard = (nblk - shard + ABD_TRAIN_SHARDS - 1) / ABD_TRAIN_SHARDS;
  const int n_shards = min(nblk, ABD_TRAIN_SHARDS);
  unsigned int* cnt_shard = cnt + (1 + shard) * ABD_TRAIN_CNT_STRIDE;
  if (wave == 0) {
    handoff_drain_stores();
    if (lane == 0) {
      const unsigned int old = handoff_count_in(cnt_shard);
      flag[0] = old + 1u == (unsigned int)n_in_shard ? 1 : 0;
    }
  }
  __syncthreads();
  const bool shard_last = flag[0] != 0;
  __syncthreads();
  if (!shard_last) return false;
  handoff_acquire();
  if (wave == 0) {
    if (lane < CB * ABD_NOUT) {
      const int cc = lane / ABD_NOUT, k = lane % ABD_NOUT;
      double v = 0.0;
      if ((mask >> cc) & 1u) {
        const double* col = rows + ((int64_t)cc * nblk + shard) * ABD_NOUT + k;
        for (int i0 = 0; i0 < n_in_shard; i0 += 8) {
          double q[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            q[u] = i0 + u < n_in_shard ? __hip_atomic_load(col + (int64_t)(i0 + u) * ABD_TRAIN_SHARDS * ABD_NOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) v += q[u];
        }
      }
      __hip_atomic_store(shard_rows + ((int64_t)cc * ABD_TRAIN_SHARDS + shard) * ABD_NOUT + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    handoff_drain_stores();
    if (lane == 0) {
      __hip_atomic_store(cnt_shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch that uses these counters)
      const unsigned int old = handoff_count_in(cnt);
      flag[0] = old + 1u == (unsigned int)n_shards ? 1 : 0;
    }
  }
  __syncthreads();
  const bool last = flag[0] != 0;
  __syncthreads();
  if (!last) return false;
  handoff_acquire();
  if (wave == 0 && lane < CB * ABD_NOUT) {
    const int cc = lane / ABD_NOUT, k = lane % ABD_NOUT;
    const double* col = shard_rows + (int64_t)cc * ABD_TRAIN_SHARDS * ABD_NOUT + k;
    double v = 0.0;
    for (int i0 = 0; i0 < n_shards; i0 += 8) {
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = i0 + u < n_shards ? __hip_atomic_load(col + (int64_t)(i0 + u) * ABD_NOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) v += q[u];
    }
    sm_chain[cc * ABD_TRAIN_SM + k] = v;
  }
  if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return true;
}

// The kernel's body for both of its entry points: abd_dense_kernel (EvalArgs: the chains' constants arrive in the kernel
// arguments, the sums go back to the host) and abd_train_kernel (DenseTrainArgs: a leapfrog-train launch -- the constants
// of a chain are those of the point its TrainChain holds, and the launch's last workgroup runs the chains' state machines).
template <typename R, int CB, bool GRAD, bool XC, typename ARGS>
__device__ __forceinline__ void dense_body(const ARGS& a) {
  constexpr bool TRAINK = is_train_args<ARGS>::value;
  static_assert(CB * ABD_NOUT <= 64, "the own-sum hand-off needs every partial row of the workgroup stored by wave 0");
  static_assert(!TRAINK || (GRAD && CB <= ABD_TRAIN_CB), "a train launch evaluates gradients for at most ABD_TRAIN_CB chains");
  extern __shared__ __align__(16) unsigned char smem[];
  const int G = a.G, N = a.N;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);              // [CB][2][G+1]
  double* red = reinterpret_cast<double*>(tabs + CB * 2 * tstride);  // [WAVES][ABD_NOUT]
  double* tab_e2 = red + ABD_WAVES_PER_BLOCK * ABD_NOUT;             // [ABD_EXP2_TAB] 2^(j/1024)
  double* dict_n = tab_e2 + ABD_EXP2_TAB;                            // XC: [ABD_XDICT] distinct log dilutions, N antigen ...
  double* dict_s = dict_n + ABD_XDICT;                               // ... and S antigen

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NSUB = ABD_WAVES_PER_BLOCK / CB;  // ranges per block
  const int c = wave % CB;                        // this wave's chain within the block's group
  const int sub = wave / CB;
  const int cbase = blockIdx.y * CB;
  // a train launch may carry one more workgroup than it has ranges, dispatched first: the SERVICE workgroup passes the
  // records of the unit's previous steps on to the host (a PCIe round trip behind a system-scope fence, ~3 us -- done by a
  // workgroup that also walks a range, that range would end ~3 us after all the others and the whole launch with it)
  int service = 0;
  if constexpr (TRAINK) {
    service = a.service;
    if (service && blockIdx.x == 0) {
      train_service(a, wave, lane);
      return;
    }
  }
  ChainPar p;
  bool chain_on = true;  // (wave-uniform) a train launch leaves the chains of its unit alone that do not step in it
  if constexpr (TRAINK) {
    const TrainChainArgs& tc = a.tc[c];
    chain_on = tc.action == ABD_TR_STEP;
    p.rw = nullptr;
    p.waner = tc.waner;
    p.iw = tc.iw;
    p.cnt = tc.cnt;
    p.perm_n = p.temp_n = p.rho_n = p.init_n = p.perm_s = p.rho_s = p.init_s = p.b_n = p.d_n = p.b_s = p.d_s = 0.0;
    if (chain_on) {
      // the point was left in device memory by the launch before this one on the stream (abd_terms.hpp: Transformed)
      abdi::chain_par_from_tr(p, tc.st->pt[tc.use_slot].tr);
    }
  } else {
    p = a.ch[cbase + c];
  }

  // Workgroups go to the 8 XCDs round-robin by id, and each XCD has its own L2: give every XCD one contiguous
  // eighth of the plane, so that the neighbouring ranges that re-read one lane group's packed words (and the rows
  // shared at range borders) find them in their own L2 instead of fetching them again.
  // blk = this workgroup's position in range order (a bijection of the workgroup ids for any grid size).
  ABD_STAMP(0);
  const int nblk = (int)gridDim.x - service;
  const int bid = (int)blockIdx.x - service;
  const int xcd = bid % 8, q8 = nblk / 8, rem8 = nblk % 8;
  const int blk = a.xcd_remap ? xcd * q8 + min(xcd, rem8) + bid / 8 : bid;

  // the workgroups that own the first ranges of grid row 0 first sum the previous launch's partials (EvalArgs::prev_*)
  if constexpr (!TRAINK) {
    const int n_fin = blockIdx.y == 0 ? a.prev_n_chains : 0;
    if (blk < n_fin) {
      finalize_chain<ABD_BLOCK>(a.prev_partials + (int64_t)blk * a.prev_blocks * ABD_NOUT, a.prev_blocks,
                     a.prev_out + (int64_t)blk * ABD_NOUT, reinterpret_cast<double*>(smem), tid, a.prev_tag);
      __syncthreads();  // the scratch becomes the power tables
    }
  }

  // the 2^(j/1024) table is requested first and stored last: its loads are in flight while everything else is set up
  double e2v[ABD_EXP2_TAB / ABD_BLOCK];
#pragma unroll
  for (int q = 0; q < ABD_EXP2_TAB / ABD_BLOCK; ++q) e2v[q] = a.exp2_tab[q * ABD_BLOCK + tid];
  double dv_n = 0.0, dv_s = 0.0;
  if (XC) {  // (ABD_XDICT == ABD_BLOCK: one entry of each dictionary per thread)
    static_assert(ABD_XDICT == ABD_BLOCK, "one dictionary entry per thread");
    if (tid < a.n_dict_n) dv_n = a.dict_n[tid];
    if (tid < a.n_dict_s) dv_s = a.dict_s[tid];
  }

  // this wave's range of the flattened (lane group, gap) plane: {first lane group, first gap, rows}, worked out from the
  // launch shape's split (abd_eval.hip: range_split; scalar arithmetic, nothing to wait for).  The first 16 ranges of every
  // grid row (the workgroups that may carry a fused sum; one range per workgroup only: CB == 4) are fin_rows shorter, the
  // others share the difference.  The split depends only on (grid.x, CB), never on the chains of the launch or on whether
  // a sum is actually carried, so results are bit-identical either way.
  auto range_start = [&](int r) { return r * a.rg_base + min(r, a.rg_extra) - a.rg_e_fin * min(r, a.rg_n_short); };
  auto range_of = [&](int r, int& r_lg, int& r_g0, int& r_rows) {
    const int pos = range_start(r);
    r_lg = G > 1 ? (int)__umulhi((uint32_t)pos, a.rg_g_magic) : pos;  // pos / G
    r_g0 = pos - r_lg * G;
    r_rows = range_start(r + 1) - pos;
  };
  int lg, g0, rows_left;
  range_of(blk * NSUB + sub, lg, g0, rows_left);
  if (TRAINK && !chain_on) rows_left = 0;
  // power-table entries this workgroup reads: up to the largest start gap of its ranges
  int n_entries = rows_left > 0 ? g0 + 1 : 0;
#pragma unroll
  for (int s = 0; s < NSUB; ++s)
    if (NSUB > 1 && s != sub) {
      int o_lg, o_g0, o_rows;
      range_of(blk * NSUB + s, o_lg, o_g0, o_rows);
      n_entries = max(n_entries, o_rows > 0 ? o_g0 + 1 : 0);
    }
  if (TRAINK && !chain_on) n_entries = 0;
  ABD_STAMP(1);
#ifdef ABD_STAMPS
  if (a.stamps && wave == 0 && lane == 0 && blockIdx.y == 0) a.stamps[(int64_t)blockIdx.x * 16 + 9] = (unsigned long long)g0;  // (diagnostic)
#endif

  // the first piece's memory accesses go out before the tables are built
  const uint32_t* vbase = reinterpret_cast<const uint32_t*>(a.vw);
  const uint32_t* ibase = reinterpret_cast<const uint32_t*>(p.iw);
  PieceLoads<R, XC> pl;
  int g1 = min(G, g0 + rows_left);
  int j = lg * 64 + lane;
  // lanes past the last individual (only the last lane group has any) sit their piece out: EXEC masks them, so
  // they neither load nor contribute and the residuals need no 0/1 guard factor
  uint32_t wi[ABD_SW], wv[ABD_SW];  // only a range's first piece can start inside an individual's gaps
  const bool first_inside = rows_left > 0 && j < N && g0 > 0;
  RowDesc<XC> rs_n = row_desc<R, XC>(a, 0, lg, g0), rs_s = row_desc<R, XC>(a, 1, lg, g0);
  if (rows_left > 0 && j < N) piece_issue_loads<R, XC>(a, rs_n, rs_s, ibase, vbase, p.waner, lane, j, g0, g1, pl);
  if (first_inside) state_issue_loads(ibase, vbase, 2u * (uint32_t)j, N, g0, wi, wv);
  // sum(i_raw), sum(ab_s_waner) of the chain: kept with the slot's discrete state (Bernoulli(i_raw | p) is on the RAW
  // matrix, abd.py:427; Q2); the first range of a chain carries them into the sums
  long long n1 = 0, m1 = 0;  // (kept as integers in scalar registers until the walk is over)
  if (blk == 0 && sub == 0 && chain_on) {
    n1 = p.cnt[0];
    m1 = p.cnt[1];
  }

  // one wave per table: with CB = 4 every wave fills its chain's two tables (one range per workgroup: entries up to its
  // start gap), with CB = 2 one each, with CB = 1 waves 0 and 1
  if (n_entries > 1) {
    if (NSUB >= 2) {
      if (sub == 0) fill_pow_table_wave(tabs + (c * 2 + 0) * tstride, p.rho_n, n_entries, lane);
      if (sub == 1) fill_pow_table_wave(tabs + (c * 2 + 1) * tstride, p.rho_s, n_entries, lane);
    } else {
      fill_pow_table_wave(tabs + (c * 2 + 0) * tstride, p.rho_n, n_entries, lane);
      fill_pow_table_wave(tabs + (c * 2 + 1) * tstride, p.rho_s, n_entries, lane);
    }
  }
  ABD_STAMP(2);
#pragma unroll
  for (int q = 0; q < ABD_EXP2_TAB / ABD_BLOCK; ++q) tab_e2[q * ABD_BLOCK + tid] = e2v[q];
  if (XC) {
    dict_n[tid] = dv_n;
    dict_s[tid] = dv_s;
  }
  ABD_STAMP(3);

  double acc[16];
#pragma unroll
  for (int k = 0; k < ABD_NACC; ++k) acc[k] = 0.0;

  const DenseChain kc = dense_chain(p);
  const double c2v = to_vgpr(ABD_EXP2_C2);
  const double2_t* tab_n = tabs + (c * 2 + 0) * tstride;
  const double2_t* tab_s = tabs + (c * 2 + 1) * tstride;
  // With one range per workgroup-wave and chain (NSUB == 1) a wave reads only the power tables it filled itself: its start
  // state needs no workgroup barrier, only its own LDS writes to have landed; the barrier (the 2^(j/1024) table is filled by
  // all waves together) then comes after the start state, when the waves have drifted apart anyway.
  if (NSUB == 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
  ABD_STAMP(4);

  // state at the end of gap g0 - 1 of the first piece: the dense design (abd.py:258-274) summed over earlier exposures
  double tn = 0.0, dn = 0.0, ts = 0.0, ds = 0.0;  // U_n, dU_n/drho_n, U_s, dU_s/drho_j
  uint32_t cfn_hi = 0, cfs_hi = 0;                 // exposure-so-far flags (abd.py:306): high word of 0.0 / 1.0
  ABD_STAMP(5);
  if (first_inside)
    dense_start_state(wi, wv, ibase, vbase, 2u * (uint32_t)j, N, g0, tab_n, tab_s, pl.wj != 0, tn, dn, ts, ds, cfn_hi, cfs_hi);
  if (NSUB == 1) __syncthreads();
  ABD_STAMP(6);

  while (rows_left > 0) {
    // ---- one piece: lane group lg, gaps [g0, g1) ----
    rows_left -= g1 - g0;
    if (j < N)
      dense_walk<R, GRAD, XC>(a, kc, rs_n, rs_s, dict_n, dict_s, ibase, vbase, 2u * (uint32_t)j, pl, lane, g0, g1, pl.wj != 0, tn, dn, ts, ds, cfn_hi, cfs_hi, tab_e2, c2v,
                          acc);
    if (rows_left <= 0) break;
    // the range goes on at gap 0 of the next lane group, from the zero state
    ++lg;
    g0 = 0;
    g1 = min(G, rows_left);
    j = lg * 64 + lane;
    tn = dn = ts = ds = 0.0;
    cfn_hi = cfs_hi = 0;
    rs_n = row_desc<R, XC>(a, 0, lg, g0);
    rs_s = row_desc<R, XC>(a, 1, lg, g0);
    if (j < N) piece_issue_loads<R, XC>(a, rs_n, rs_s, ibase, vbase, p.waner, lane, j, g0, g1, pl);
  }

  ABD_STAMP(7);
  acc[ABD_NACC] = lane == 0 ? (double)n1 : 0.0;
  acc[ABD_NACC + 1] = lane == 0 ? (double)m1 : 0.0;
  acc[ABD_NACC + 2] = 0.0;
  // ---- reduction: lanes -> wave -> block (LDS) -> per-block partial in global memory ----
  const double tot = wave_reduce16(acc, lane);
  if ((lane & 3) == 0) red[wave * ABD_NOUT + reduce16_index(lane)] = tot;
  __syncthreads();
  bool own_sum = TRAINK;  // does the launch sum its own partial rows?
  if constexpr (!TRAINK) own_sum = a.fin_count != nullptr || a.fin_count2 != nullptr;
  if (tid < CB * ABD_NOUT) {
    const int cc = tid / ABD_NOUT, k = tid % ABD_NOUT;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < NSUB; ++w) v += red[(w * CB + cc) * ABD_NOUT + k];  // waves w*CB + cc hold chain cc
    double* dst = a.partials + ((int64_t)(cbase + cc) * nblk + blk) * ABD_NOUT + k;  // rows in range order
    if (own_sum)
      __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another workgroup of THIS launch
    else
      *dst = v;
  }
  ABD_STAMP(8);
  if (!own_sum) return;

  // ---- own fixed-order sum: the workgroup that counts in last sums the partial rows itself instead of a second launch.
  // Hand-off (abd_device.hpp: handoff_count_in): every partial row of this workgroup was stored write-through (sc1) by wave 0
  // (CB x 16 <= 64 lanes) in front of a workgroup barrier; one lane per counter counts in with a returning agent-scope
  // acq_rel add; the workgroup whose add came last re-reads the rows with sc1 loads behind another barrier
  int* flag = reinterpret_cast<int*>(red);  // the block reduction is done with: [CB] flags
  __syncthreads();
  if constexpr (TRAINK) {
    // Count-in in two levels: with one counter a launch of 1 024 workgroups spends ~12 us in its 1 024 returning adds (they
    // serialise, ~12 ns each), and its last workgroup would read 128 KB of rows per chain by itself.  Workgroup b belongs to
    // shard b mod ABD_TRAIN_SHARDS; a shard's last arriver sums the shard's rows of every stepping chain (in range order) into
    // one shard row and counts in at the top; the top's last arriver sums the shard rows (in shard order) and runs the
    // chains' state machines, wave k chain k's (abd_train.hpp).  Fixed orders: the sums depend on the launch shape only.
    // Hand-off as above: rows stored write-through by wave 0, drained, then the returning agent-scope add; re-read with sc1 loads.
    // (a launch of at most ABD_TRAIN_ONE_LEVEL workgroups -- one per CU: a unit of one chain -- counts in at the top directly and
    // its last workgroup sums all rows, 256 threads wide: one hop less on the path every leapfrog of the chain waits for)
    if (nblk <= ABD_TRAIN_ONE_LEVEL) {
      if (wave == 0) {
        handoff_drain_stores();
        if (lane == 0) {
          const unsigned int old = handoff_count_in(a.fin_count);
          flag[0] = old + 1u == (unsigned int)nblk ? 1 : 0;
        }
      }
      __syncthreads();
      const bool last1 = flag[0] != 0;
      __syncthreads();
      if (!last1) return;
      ABD_STAMP(10);
      handoff_acquire();
      double* sm = reinterpret_cast<double*>(smem);
      double* sm_chain1 = sm + ABD_FIN_PARTS * ABD_NOUT;  // [CB][ABD_TRAIN_SM] behind the sum's scratch
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        if (a.tc[cc].action == ABD_TR_STEP) {  // (workgroup-uniform)
          sum_chain_coherent<ABD_BLOCK>(a.partials + (int64_t)cc * nblk * ABD_NOUT, nblk, sm, tid);
          if (tid < ABD_NOUT) sm_chain1[cc * ABD_TRAIN_SM + tid] = sm[tid];
          __syncthreads();
        }
      }
      ABD_STAMP(11);
      if (wave < CB && a.tc[wave].action != ABD_TR_SKIP) train_step(a, a.tc[wave], sm_chain1 + wave * ABD_TRAIN_SM, lane);
      ABD_STAMP(12);
      if (tid == 0) __hip_atomic_store(a.fin_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    unsigned int step_mask = 0;
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) step_mask |= a.tc[cc].action == ABD_TR_STEP ? 1u << cc : 0u;
    double* sm_chain = reinterpret_cast<double*>(smem);  // [CB][ABD_TRAIN_SM]
    if (!two_level_sums<CB>(a.partials, a.partials + (int64_t)CB * nblk * ABD_NOUT, a.fin_count, nblk, blk, step_mask, flag, sm_chain, tid)) return;
    if (wave < CB && a.tc[wave].action != ABD_TR_SKIP) train_step(a, a.tc[wave], sm_chain + wave * ABD_TRAIN_SM, lane);
  } else if (a.fin_count2 != nullptr) {
    // a synchronous call's launch (a grid that fills the chip: 1 024 workgroups): counted in in two levels like a train launch;
    // the workgroup that comes last writes each chain's row -- 15 sums, the tag behind a system-scope fence -- for the host
    double* sm_chain = reinterpret_cast<double*>(smem);
    const int n_rows_chains = (int)gridDim.y * CB;  // chains of the launch: the shard rows lie behind all partial rows
    if (!two_level_sums<CB>(a.partials + (int64_t)cbase * nblk * ABD_NOUT,
                            a.partials + ((int64_t)n_rows_chains * nblk + (int64_t)cbase * ABD_TRAIN_SHARDS) * ABD_NOUT,
                            a.fin_count2 + (int64_t)blockIdx.y * (1 + ABD_TRAIN_SHARDS) * ABD_TRAIN_CNT_STRIDE, nblk, blk, (1u << CB) - 1u, flag,
                            sm_chain, tid))
      return;
    if (wave == 0) {
      if (lane < CB * ABD_NOUT && lane % ABD_NOUT < ABD_NOUT - 1)
        a.fin_out[(int64_t)(cbase + lane / ABD_NOUT) * ABD_NOUT + lane % ABD_NOUT] = sm_chain[(lane / ABD_NOUT) * ABD_TRAIN_SM + lane % ABD_NOUT];
      __threadfence_system();
      __builtin_amdgcn_wave_barrier();
      if (lane < CB) __hip_atomic_store(a.fin_out + (int64_t)(cbase + lane) * ABD_NOUT + (ABD_NOUT - 1), a.fin_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  } else {
    if (wave == 0) {
      handoff_drain_stores();
      if (lane < CB) {
        const unsigned int old = handoff_count_in(a.fin_count + cbase + lane);
        flag[lane] = old + 1u == gridDim.x ? 1 : 0;
      }
    }
    __syncthreads();
    bool last[CB];  // workgroup-uniform; read before the sum's scratch may overwrite the flags
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) last[cc] = flag[cc] != 0;
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      if (last[cc]) {
        handoff_acquire();
        finalize_chain_coherent<ABD_BLOCK>(a.partials + (int64_t)(cbase + cc) * gridDim.x * ABD_NOUT, (int)gridDim.x,
                                           a.fin_out + (int64_t)(cbase + cc) * ABD_NOUT, reinterpret_cast<double*>(smem), tid, a.fin_tag);
        if (tid == 0) __hip_atomic_store(a.fin_count + cbase + cc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
      }
    }
  }
}

template <typename R, int CB, bool GRAD, bool XC>
#ifndef ABD_DENSE_MINW
#define ABD_DENSE_MINW 4
#endif
__global__ __launch_bounds__(ABD_BLOCK, ABD_DENSE_MINW) void abd_dense_kernel(const EvalArgs a) {  // 4 waves per SIMD: <= 128 VGPRs
  dense_body<R, CB, GRAD, XC, EvalArgs>(a);
}

// a leapfrog-train launch of a unit of CB chains (abd_train.hpp; abd_sampler.hip)
template <typename R, int CB, bool XC>
__global__ __launch_bounds__(ABD_BLOCK, ABD_DENSE_MINW) void abd_train_kernel(const DenseTrainArgs a) {
  dense_body<R, CB, true, XC, DenseTrainArgs>(a);
}
