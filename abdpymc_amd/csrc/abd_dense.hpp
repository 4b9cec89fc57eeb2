// abd_dense.hpp -- dense-panel evaluation kernel (included by abd_kernels.hpp after the shared helpers).
//
// lane = individual.  The (lane group, gap) plane -- n_lg groups of 64 individuals x G gap rows -- is cut
// into equal contiguous ranges, one per wave "slot", so every slot walks the same number of gap rows
// (+-1) whatever N and G are, and the grid is an exact multiple of the CU count.  A range is walked as
// one or two pieces (it may end one lane group and begin the next); a piece that does not start at gap 0
// rebuilds its start state per lane from the packed words: constrain on the words (abd.py:640-667), then
// the reference's dense design (abd.py:258-274) summed over the set bits before the piece with rho^k read
// from the chain's LDS table.  Inside a piece the responses advance by the recurrence (abd.py:288).
//
// The 4 waves of a workgroup are CB chains x (4 / CB) neighbouring ranges: with CB = 4 the panel rows
// they share are fetched from HBM once and served from L1/L2 to the other three.
//
// The kernel is bound by vector-instruction issue (fp64 instructions issue at half the rate of 32-bit ones),
// so the gap loop is written for instruction count:
//   * indicator bits -> 0.0 / 1.0 doubles by v_bfe_i32 (scalar bit index) + v_and with the high word of 1.0;
//     the "exposed so far" flags of perm_response (abd.py:306) are integer ORs of those high words
//   * e^u = 2^t with 1024 t = 1024 e + j + f: T[j] = 2^(j/1024) from a 1024-entry LDS table, a cubic in f,
//     the exponent e added into T's high word; 1 + 2^t is ONE fma (tools/exp2_table.py: 3.5e-16)
//   * one v_rcp_f64 per cell for both antigens (1/A = B/(AB))
//   * gap rows are addressed as scalar row base + a constant per-lane offset (no vector address arithmetic)
#pragma once

#include "abd_device.hpp"


// bits [g0, g0 + 32) of the packed row, moved to bit 0 (g0 wave-uniform; bits past the row's end read as zero).
// Every word is visited with a compile-time index (a runtime-indexed register array would go to scratch); the
// uniform comparisons are scalar branches, the work is one v_alignbit_b32.
__device__ __forceinline__ uint32_t extract_bits32(const uint64_t w[ABD_MAXT], int g0) {
  const int idx = g0 >> 5;
  const uint32_t sh = (uint32_t)g0 & 31u;
  uint32_t v = 0;
#pragma unroll
  for (int q = 0; q < 2 * ABD_MAXT; ++q) {
    if (q == idx) {
      const uint32_t lo = (q & 1) ? (uint32_t)(w[q >> 1] >> 32) : (uint32_t)w[q >> 1];
      const uint32_t hi = q + 1 < 2 * ABD_MAXT ? (((q + 1) & 1) ? (uint32_t)(w[(q + 1) >> 1] >> 32) : (uint32_t)w[(q + 1) >> 1]) : 0u;
      v = __builtin_amdgcn_alignbit(hi, lo, sh);
    }
  }
  return v;
}

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


// Both antigens of one cell at once: the two reciprocals 1/(1+e_n), 1/(1+e_s) come from ONE v_rcp_f64 (quarter
// rate) of the product -- 1/A = B/(AB), 1/B = A/(AB).  c_n = b_n log2(e) 1024, c_s likewise (wave-uniform).
template <bool GRAD>
__device__ __forceinline__ void obs_pair_tab(double an, double xn, double yn, double c_n, double d_n, double as, double xs,
                                             double ys, double c_s, double d_s, const double* tab, double (&acc)[16],
                                             double& h_n, double& h_s) {
  const double amx_n = an - xn, amx_s = as - xs;
  const double A = one_plus_exp2_tab(c_n * amx_n, tab);
  const double B = one_plus_exp2_tab(c_s * amx_s, tab);
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

// The chain's constants as the gap loop wants them (wave-uniform; a VOP3 fma reads at most one scalar operand, so the
// addends that meet another chain constant live in VGPRs, or every use costs a v_mov_b64).
struct DenseChain {
  double rho_n, temp_n, rho_s, init_n, init_s, perm_n, perm_s, d_n, d_s, c_n, c_s;
};
__device__ __forceinline__ DenseChain dense_chain(double perm_n, double temp_n, double rho_n, double init_n, double perm_s,
                                                  double rho_s, double init_s, double b_n, double d_n, double b_s, double d_s) {
  DenseChain k;
  k.rho_n = rho_n;
  k.temp_n = temp_n;
  k.rho_s = rho_s;
  k.init_n = to_vgpr(init_n);
  k.init_s = to_vgpr(init_s);
  k.perm_n = perm_n;
  k.perm_s = perm_s;
  k.d_n = d_n;
  k.d_s = d_s;
  k.c_n = b_n * (1.4426950408889634074 * ABD_EXP2_TAB);
  k.c_s = b_s * (1.4426950408889634074 * ABD_EXP2_TAB);
  return k;
}

// State of one lane's individual at the end of gap g0 - 1 (g0 > 0): the dense design (abd.py:258-274) summed over the
// exposures before the piece, rho^k from the chain's LDS tables (tab_s: the waning table or the table of ones).
__device__ __forceinline__ void dense_start_state(const uint64_t (&I)[ABD_MAXT], const uint64_t (&V)[ABD_MAXT], int g0,
                                                  const double2_t* tab_n, const double2_t* tab_s, double& tn, double& dn,
                                                  double& ts, double& ds, uint32_t& cfn_hi, uint32_t& cfs_hi) {
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    if (t * 64 < g0) {
      const int rel = g0 - t * 64;  // bits < rel of word t are before the piece
      const uint64_t below = rel >= 64 ? ~0ull : ((1ull << rel) - 1ull);
      uint64_t mi = I[t] & below, mv = V[t] & below;
      if (mi != 0) cfn_hi = 0x3FF00000u;
      if ((mi | mv) != 0) cfs_hi = 0x3FF00000u;
      while (mi) {  // per-lane trip count
        const int b = __builtin_ctzll(mi);
        mi &= mi - 1;
        const int idx = g0 - (t * 64 + b);  // = k + 1 with k = (g0 - 1) - r
        const double2_t pn = tab_n[idx];
        const double2_t ps = tab_s[idx];
        tn += pn.x;
        dn += pn.y;
        ts += ps.x;
        ds += ps.y;
      }
      while (mv) {
        const int b = __builtin_ctzll(mv);
        mv &= mv - 1;
        const double2_t ps = tab_s[g0 - (t * 64 + b)];
        ts += ps.x;
        ds += ps.y;
      }
    }
  }
}

// Walk gaps [g0, g1) of lane group lg: recurrence form (abd.py:288) + likelihood terms into acc.
template <typename R, bool GRAD>
__device__ __forceinline__ void dense_walk(const EvalArgs& a, const DenseChain& k, const uint64_t (&I)[ABD_MAXT],
                                           const uint64_t (&V)[ABD_MAXT], int lg, int lane, int g0, int g1, bool wj, double tn,
                                           double dn, double ts, double ds, uint32_t cfn_hi, uint32_t cfs_hi,
                                           const double* tab_e2, double (&acc)[16]) {
  const int N = a.N;
  const double rho_n = k.rho_n, temp_n = k.temp_n, init_n = k.init_n, init_s = k.init_s, perm_n = k.perm_n, perm_s = k.perm_s;
  const double d_n = k.d_n, d_s = k.d_s, c_n = k.c_n, c_s = k.c_s;
  const double rho_j = wj ? k.rho_s : 1.0;  // abd.py:374
  double hd_s = 0.0;
  const uint32_t lane_off = (uint32_t)lane * (uint32_t)sizeof(YX<R>);
  const uint32_t rstride = (uint32_t)N * (uint32_t)sizeof(YX<R>);  // 34 rows of it fit 32 bits (abd_create checks)
  const uint32_t z_ei = zero_vgpr(), z_ev = zero_vgpr(), z_cn = zero_vgpr(), z_cs = zero_vgpr();
  for (int gc = g0; gc < g1; gc += 32) {  // <= 32 gaps of indicator bits at a time
    const int len = min(32, g1 - gc);
    const uint32_t seg_i = extract_bits32(I, gc);
    const uint32_t seg_v = extract_bits32(V, gc);
    // gap rows through buffer loads: descriptor base = this chunk's first row of this lane group (scalar), scalar
    // offset = row within the chunk, vector offset = the lane's constant -- no vector address arithmetic at all
    const int64_t row0 = ((int64_t)gc * N + (int64_t)lg * 64) * (int64_t)sizeof(YX<R>);
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.yx_n)) + row0, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.yx_s)) + row0, 0, -1, 0x00020000);
    auto ldrow = [&](const __amdgpu_buffer_rsrc_t& rs, int gi) { return load_yx<R>(rs, lane_off, (uint32_t)gi * rstride); };

    // one gap: 0/1 indicators enter as doubles, so the perm switch (abd.py:306) is an fma, not a select
    auto step = [&](int gi, const YX<R>& on, const YX<R>& os) {
      const uint32_t ei_hi = (uint32_t)__builtin_amdgcn_sbfe((int)seg_i, (uint32_t)gi, 1u) & 0x3FF00000u;
      const uint32_t ev_hi = (uint32_t)__builtin_amdgcn_sbfe((int)seg_v, (uint32_t)gi, 1u) & 0x3FF00000u;
      const double e_i = hi_to_double(ei_hi, z_ei), e_v = hi_to_double(ev_hi, z_ev);
      cfn_hi |= ei_hi;
      cfs_hi |= ei_hi | ev_hi;
      dn = fma_s(rho_n, dn, tn);
      tn = fma_s(rho_n, tn, e_i);
      ds = fma_v(rho_j, ds, ts);
      ts = fma_v(rho_j, ts, e_i + e_v);  // unit boosts: temp unused (abd.py:272)
      const double cf_n = hi_to_double(cfn_hi, z_cn), cf_s = hi_to_double(cfs_hi, z_cs);
      // mu_n = perm + temp + init   abd.py:341 ; mu_s = perm + tinf + tvac + init   abd.py:389-391
      const double an = fma(temp_n, tn, fma(cf_n, perm_n, init_n));
      const double as = fma(cf_s, perm_s, init_s) + ts;
      double h_n = 0.0, h_s = 0.0;
      obs_pair_tab<GRAD>(an, (double)on.x, (double)on.y, c_n, d_n, as, (double)os.x, (double)os.y, c_s, d_s, tab_e2, acc,
                         h_n, h_s);
      if (GRAD) {
        acc[A_N_HC] = fma(h_n, cf_n, acc[A_N_HC]);
        acc[A_N_HU] = fma(h_n, tn, acc[A_N_HU]);
        acc[A_N_HD] = fma(h_n, dn, acc[A_N_HD]);
        acc[A_S_HC] = fma(h_s, cf_s, acc[A_S_HC]);
        hd_s = fma(h_s, ds, hd_s);
      }
    };

    // two row buffers; each is refilled right after the step that consumed it, for the step two gaps on
    const int last = len - 1;
    YX<R> n0 = ldrow(rs_n, 0), s0 = ldrow(rs_s, 0);
    YX<R> n1 = ldrow(rs_n, min(1, last)), s1 = ldrow(rs_s, min(1, last));
    int gi = 0;
    for (; gi + 1 < len; gi += 2) {
      const int ga = min(gi + 2, last), gb = min(gi + 3, last);
      step(gi, n0, s0);
      n0 = ldrow(rs_n, ga);
      s0 = ldrow(rs_s, ga);
      step(gi + 1, n1, s1);
      n1 = ldrow(rs_n, gb);
      s1 = ldrow(rs_s, gb);
    }
    if (gi < len) step(gi, n0, s0);
  }
  acc[A_S_HD] += wj ? hd_s : 0.0;  // d rho_j / d rho_s = waner_j
}

template <typename R, int CB, bool GRAD>
__global__ __launch_bounds__(ABD_BLOCK, 4) void abd_dense_kernel(const EvalArgs a) {  // 4 waves per SIMD: <= 128 VGPRs
  // LDS: [CB][2][G+1] power tables, [G+1] ones table, block reduction, 2^(j/1024) table
  extern __shared__ __align__(16) unsigned char smem[];
  const int G = a.G, N = a.N, nt = a.nt;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + CB * 2 * tstride;
  double* red = reinterpret_cast<double*>(tab_ones + tstride);  // [WAVES][ABD_NOUT]
  double* tab_e2 = red + ABD_WAVES_PER_BLOCK * ABD_NOUT;           // [ABD_EXP2_TAB] 2^(j/1024)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NSUB = ABD_WAVES_PER_BLOCK / CB;  // ranges per block
  const int c = wave % CB;                        // this wave's chain within the block's group
  const int sub = wave / CB;
  const int cbase = blockIdx.y * CB;
  const ChainPar& p = a.ch[cbase + c];

  // Workgroups go to the 8 XCDs round-robin by id, and each XCD has its own L2: give every XCD one contiguous
  // eighth of the plane, so that the ~G/rows-per-range neighbouring ranges that re-read one lane group's packed
  // words (and the rows shared at range borders) find them in their own L2 instead of fetching them again.
  // blk = this workgroup's position in range order (a bijection of blockIdx.x for any grid size).
  ABD_STAMP(0);
  const int nblk = (int)gridDim.x;
  const int xcd = (int)blockIdx.x % 8, q8 = nblk / 8, rem8 = nblk % 8;
  const int blk = a.xcd_remap ? xcd * q8 + min(xcd, rem8) + (int)blockIdx.x / 8 : (int)blockIdx.x;

  // the workgroups that own the first ranges of grid row 0 first sum the previous launch's partials (EvalArgs::prev_*)
  const int n_fin = blockIdx.y == 0 ? a.prev_n_chains : 0;
  if (blk < n_fin) {
    finalize_chain<ABD_BLOCK>(a.prev_partials + (int64_t)blk * a.prev_blocks * ABD_NOUT, a.prev_blocks,
                   a.prev_out + (int64_t)blk * ABD_NOUT, reinterpret_cast<double*>(smem), tid, a.prev_tag);
    __syncthreads();  // the scratch becomes the power tables
  }

  // this wave's range of the flattened (lane group, gap) plane: {first lane group, first gap, rows} from the table the
  // host built for this launch shape (abd_capi.hip: range_table) -- the 64-bit divisions that cut the plane into equal
  // ranges cost ~700 scalar and ~80 vector instructions per wave when done here.  The first 16 ranges of every grid row
  // (the workgroups that may carry a fused sum; one range per workgroup only: CB == 4) are fin_rows shorter, the
  // others share the difference.  The split depends only on (grid.x, CB), never on the chains of the launch or on
  // whether a sum is actually carried, so results are bit-identical either way.
  const int r = blk * NSUB + sub;
  const int4 rt = reinterpret_cast<const int4*>(a.range_tab)[r];
  int lg = rt.x, g0 = rt.y, rows_left = rt.z;
  const bool has_work = rows_left > 0;
  const int g0_first = has_work ? g0 : 0;

  ABD_STAMP(1);
  // one wave per chain fills that chain's two power tables, the last wave the ones table.  Only a piece
  // that starts inside an individual's gaps reads them, and only entries up to its start gap.
  const int n_entries = CB == ABD_WAVES_PER_BLOCK ? g0_first + 1 : tstride;
  // the 2^(j/1024) table is requested first and stored last: its loads are in flight while the power tables are built
  double e2v[ABD_EXP2_TAB / ABD_BLOCK];
#pragma unroll
  for (int q = 0; q < ABD_EXP2_TAB / ABD_BLOCK; ++q) e2v[q] = a.exp2_tab[q * ABD_BLOCK + tid];
  if (NSUB >= 2) {  // a chain has two or four waves in this workgroup: one fills its rho_n table, another its rho_s table
    if (sub == 0) fill_pow_table_wave(tabs + (c * 2 + 0) * tstride, p.rho_n, n_entries, lane);
    if (sub == 1) fill_pow_table_wave(tabs + (c * 2 + 1) * tstride, p.rho_s, n_entries, lane);
  } else {
    fill_pow_table_wave(tabs + (c * 2 + 0) * tstride, p.rho_n, n_entries, lane);
    fill_pow_table_wave(tabs + (c * 2 + 1) * tstride, p.rho_s, n_entries, lane);
  }
  if (wave == ABD_WAVES_PER_BLOCK - 1) fill_ones_table_wave(tab_ones, n_entries, lane);
  ABD_STAMP(2);
#pragma unroll
  for (int q = 0; q < ABD_EXP2_TAB / ABD_BLOCK; ++q) tab_e2[q * ABD_BLOCK + tid] = e2v[q];
  ABD_STAMP(3);

  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;

  const DenseChain kc = dense_chain(p.perm_n, p.temp_n, p.rho_n, p.init_n, p.perm_s, p.rho_s, p.init_s, p.b_n, p.d_n, p.b_s, p.d_s);
  const double2_t* tab_n = tabs + (c * 2 + 0) * tstride;
  const double2_t* tab_sw = tabs + (c * 2 + 1) * tstride;
  __syncthreads();
  ABD_STAMP(4);

  for (; rows_left > 0; ++lg, g0 = 0) {
    // ---- one piece: lane group lg, gaps [g0, g1) ----
    const int g1 = min(G, g0 + rows_left);
    rows_left -= g1 - g0;
    // lanes past the last individual (only the last lane group has any) sit the piece out: EXEC masks them, so
    // they neither load nor contribute and the residuals need no 0/1 guard factor
    const int j = lg * 64 + lane;
    if (j < N) {

    // packed indicator rows of this lane's individual; constrain (abd.py:640-667)
    uint64_t V[ABD_MAXT], P[ABD_MAXT], Rw[ABD_MAXT], I[ABD_MAXT];
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) {
      V[t] = P[t] = Rw[t] = 0;
      if (t < nt) {
        V[t] = a.vw[(int64_t)t * N + j];
        if (a.pw) P[t] = a.pw[(int64_t)t * N + j];
        Rw[t] = p.rw[(int64_t)t * N + j];
      }
    }
    const bool wj = p.waner[j] != 0;
    constrain_masks(Rw, P, a, I);
#ifdef ABD_STAMPS
    if (I[0] == 0x123456789abcdefull) acc[15] += 1.0;  // keeps the stamp behind the loads
#endif
    ABD_STAMP(5);
    if (g0 == 0) {  // each individual's gap 0 belongs to exactly one piece
      int n1 = 0;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) n1 += __builtin_popcountll(Rw[t]);  // Bernoulli(i_raw | p) is on the RAW matrix
      acc[ABD_NACC] += (double)n1;
      acc[ABD_NACC + 1] += wj ? 1.0 : 0.0;
    }

    // state at the end of gap g0 - 1: the dense design (abd.py:258-274) summed over earlier exposures
    double tn = 0.0, dn = 0.0, ts = 0.0, ds = 0.0;  // U_n, dU_n/drho_n, U_s, dU_s/drho_j
    uint32_t cfn_hi = 0, cfs_hi = 0;                 // exposure-so-far flags (abd.py:306): high word of 0.0 / 1.0
    if (g0 > 0) dense_start_state(I, V, g0, tab_n, wj ? tab_sw : tab_ones, tn, dn, ts, ds, cfn_hi, cfs_hi);

    ABD_STAMP(6);
    dense_walk<R, GRAD>(a, kc, I, V, lg, lane, g0, g1, wj, tn, dn, ts, ds, cfn_hi, cfs_hi, tab_e2, acc);
    }  // j < N
  }

  ABD_STAMP(7);
  // ---- reduction: lanes -> wave -> block (LDS) -> per-block partial in global memory ----
  const double tot = wave_reduce16(acc, lane);
  if ((lane & 3) == 0) red[wave * ABD_NOUT + reduce16_index(lane)] = tot;
  __syncthreads();
  if (tid < CB * ABD_NOUT) {
    const int cc = tid / ABD_NOUT, k = tid % ABD_NOUT;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < NSUB; ++w) v += red[(w * CB + cc) * ABD_NOUT + k];  // waves w*CB + cc hold chain cc
    double* dst = a.partials + ((int64_t)(cbase + cc) * gridDim.x + blk) * ABD_NOUT + k;  // rows in range order
    if (a.fin_count)
      __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another workgroup of THIS launch
    else
      *dst = v;
  }
  ABD_STAMP(8);
  if (!a.fin_count) return;

  // ---- own fixed-order sum (a sampler unit's launch, ABD_DENSE_OWN_SUM=1): the workgroup that counts in last for a chain
  // sums that chain's partial rows itself instead of a second launch; hand-off and order as in abd_obs_kernel ----
  int* flag = reinterpret_cast<int*>(red);  // the block reduction is done with: [CB] flags
  __syncthreads();
  if (wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < CB) {
      const unsigned int old = __hip_atomic_fetch_add(a.fin_count + cbase + lane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      finalize_chain_coherent<ABD_BLOCK>(a.partials + (int64_t)(cbase + cc) * gridDim.x * ABD_NOUT, (int)gridDim.x,
                                         a.fin_out + (int64_t)(cbase + cc) * ABD_NOUT, reinterpret_cast<double*>(smem), tid, a.fin_tag);
      if (tid == 0) __hip_atomic_store(a.fin_count + cbase + cc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
  }
}
