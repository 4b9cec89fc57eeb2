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
#pragma once

// bits [g0, g0 + len) of the packed row, moved to bit 0 (len <= 64; g0 wave-uniform).  Every word is
// visited with a compile-time index (a runtime-indexed register array would go to scratch).
__device__ __forceinline__ uint64_t extract_bits(const uint64_t w[ABD_MAXT], int g0, int len) {
  uint64_t v = 0;
#pragma unroll
  for (int t = 0; t < ABD_MAXT; ++t) {
    const int sh = g0 - t * 64;  // wave-uniform
    if (sh >= 0 && sh < 64) v |= w[t] >> sh;
    if (sh < 0 && sh > -64) v |= w[t] << (-sh);
  }
  return len >= 64 ? v : (v & ((1ull << len) - 1ull));
}

template <typename R, int CB, bool GRAD>
__global__ __launch_bounds__(ABD_BLOCK, 4) void abd_dense_kernel(const EvalArgs a) {  // 4 waves per SIMD: <= 128 VGPRs
  // LDS: [CB][2][G+1] power tables, [G+1] ones table, block reduction
  extern __shared__ __align__(16) unsigned char smem[];
  const int G = a.G, N = a.N, nt = a.nt;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + CB * 2 * tstride;
  double* red = reinterpret_cast<double*>(tab_ones + tstride);  // [WAVES][ABD_NOUT]

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

  // this wave's range of the flattened (lane group, gap) plane.  The workgroups that may carry a fused sum
  // (the first n_chains of grid row 0; one range per workgroup only: CB == 4) get ranges fin_rows shorter,
  // the others share the difference.  The split depends only on the launch shape, never on whether a sum is
  // actually carried, so results are bit-identical either way.
  const int64_t rows_total = (int64_t)a.n_lg * G;
  const int64_t n_ranges = (int64_t)gridDim.x * NSUB;
  const int64_t r = (int64_t)blk * NSUB + sub;
  const int64_t n_short = blockIdx.y == 0 ? min((int64_t)a.n_chains, n_ranges) : 0;
  const int64_t e_fin = (NSUB == 1 && (rows_total + n_short * a.fin_rows) / n_ranges >= 2 * a.fin_rows) ? a.fin_rows : 0;
  const int64_t virt = rows_total + n_short * e_fin;
  int64_t pos = r * virt / n_ranges - e_fin * min(r, n_short);
  const int64_t end = (r + 1) * virt / n_ranges - e_fin * min(r + 1, n_short);

  const bool has_work = pos < end;
  const int g0_first = has_work ? (int)(pos % G) : 0;

  // one wave per chain fills that chain's two power tables, the last wave the ones table.  Only a piece
  // that starts inside an individual's gaps reads them, and only entries up to its start gap.
  const int n_entries = CB == ABD_WAVES_PER_BLOCK ? g0_first + 1 : tstride;
  if (sub == 0) {
    fill_pow_table_wave(tabs + (c * 2 + 0) * tstride, p.rho_n, n_entries, lane);
    fill_pow_table_wave(tabs + (c * 2 + 1) * tstride, p.rho_s, n_entries, lane);
  }
  if (wave == ABD_WAVES_PER_BLOCK - 1) fill_ones_table_wave(tab_ones, n_entries, lane);

  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;

  const double rho_n = p.rho_n, temp_n = p.temp_n, rho_s = p.rho_s;
  // a VOP3 fma reads at most one scalar operand: keep the addends that meet another chain constant in
  // VGPRs, or every use costs a v_mov_b64
  const double init_n = to_vgpr(p.init_n), init_s = to_vgpr(p.init_s);
  const double perm_n = p.perm_n, perm_s = p.perm_s;
  const double b_n = p.b_n, d_n = p.d_n, b_s = p.b_s, d_s = p.d_s;
  const double2_t* tab_n = tabs + (c * 2 + 0) * tstride;
  const double2_t* tab_sw = tabs + (c * 2 + 1) * tstride;
  __syncthreads();

  while (pos < end) {
    // ---- one piece: lane group lg, gaps [g0, g1) ----
    const int lg = (int)(pos / G);
    const int g0 = (int)(pos - (int64_t)lg * G);
    const int g1 = (int)min((int64_t)G, (int64_t)g0 + (end - pos));
    pos += g1 - g0;
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
    if (g0 == 0) {  // each individual's gap 0 belongs to exactly one piece
      int n1 = 0;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) n1 += __builtin_popcountll(Rw[t]);  // Bernoulli(i_raw | p) is on the RAW matrix
      acc[ABD_NACC] += (double)n1;
      acc[ABD_NACC + 1] += wj ? 1.0 : 0.0;
    }

    // state at the end of gap g0 - 1: the dense design (abd.py:258-274) summed over earlier exposures
    double tn = 0.0, dn = 0.0, ts = 0.0, ds = 0.0;  // U_n, dU_n/drho_n, U_s, dU_s/drho_j
    double cf_n = 0.0, cf_s = 0.0;                   // exposure-so-far flags as 0.0 / 1.0 (abd.py:306)
    if (g0 > 0) {
      const double2_t* tab_s = wj ? tab_sw : tab_ones;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) {
        if (t * 64 < g0) {
          const int rel = g0 - t * 64;  // bits < rel of word t are before the piece
          const uint64_t below = rel >= 64 ? ~0ull : ((1ull << rel) - 1ull);
          uint64_t mi = I[t] & below, mv = V[t] & below;
          if (mi != 0) cf_n = 1.0;
          if ((mi | mv) != 0) cf_s = 1.0;
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

    // ---- walk the piece: recurrence form (abd.py:288) + likelihood terms ----
    const double rho_j = wj ? rho_s : 1.0;  // abd.py:374
    double hd_s = 0.0;
    for (int gc = g0; gc < g1; gc += 64) {  // <= 64 gaps of indicator bits at a time
      const int len = min(64, g1 - gc);
      uint64_t seg_i = extract_bits(I, gc, len);
      uint64_t seg_v = extract_bits(V, gc, len);
      // gap rows: wave-uniform base + this lane's individual
      const YX<R>* row_n = reinterpret_cast<const YX<R>*>(a.yx_n) + (int64_t)gc * N;
      const YX<R>* row_s = reinterpret_cast<const YX<R>*>(a.yx_s) + (int64_t)gc * N;

      // one gap: 0/1 indicators enter as doubles, so the perm switch (abd.py:306) is an fma, not a select
      auto step = [&](const YX<R>& on, const YX<R>& os) {
        const uint32_t ib = (uint32_t)seg_i & 1u, vb = (uint32_t)seg_v & 1u;
        seg_i >>= 1;
        seg_v >>= 1;
        const double e_i = (double)ib, e_v = (double)vb;
        dn = fma(rho_n, dn, tn);
        tn = fma(rho_n, tn, e_i);
        ds = fma(rho_j, ds, ts);
        ts = fma(rho_j, ts, e_i + e_v);  // unit boosts: temp unused (abd.py:272)
        cf_n = fmax(cf_n, e_i);
        cf_s = fmax(cf_s, fmax(e_i, e_v));
        // mu_n = perm + temp + init   abd.py:341 ; mu_s = perm + tinf + tvac + init   abd.py:389-391
        const double an = fma(temp_n, tn, fma(cf_n, perm_n, init_n));
        const double as = fma(cf_s, perm_s, init_s) + ts;
        double h_n = 0.0, h_s = 0.0;
        obs_pair<GRAD>(an, (double)on.x, (double)on.y, b_n, d_n, as, (double)os.x, (double)os.y, b_s, d_s, acc, h_n, h_s);
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
      YX<R> n0 = row_n[j], s0 = row_s[j];
      YX<R> n1 = row_n[(int64_t)min(1, last) * N + j], s1 = row_s[(int64_t)min(1, last) * N + j];
      int gi = 0;
      for (; gi + 1 < len; gi += 2) {
        const int ga = min(gi + 2, last), gb = min(gi + 3, last);
        step(n0, s0);
        n0 = row_n[(int64_t)ga * N + j];
        s0 = row_s[(int64_t)ga * N + j];
        step(n1, s1);
        n1 = row_n[(int64_t)gb * N + j];
        s1 = row_s[(int64_t)gb * N + j];
      }
      if (gi < len) step(n0, s0);
    }
    acc[A_S_HD] += wj ? hd_s : 0.0;  // d rho_j / d rho_s = waner_j
    }  // j < N
  }

  // ---- reduction: lanes -> wave -> block (LDS) -> per-block partial in global memory ----
  const double tot = wave_reduce16(acc, lane);
  if ((lane & 3) == 0) red[wave * ABD_NOUT + reduce16_index(lane)] = tot;
  __syncthreads();
  if (tid < CB * ABD_NOUT) {
    const int cc = tid / ABD_NOUT, k = tid % ABD_NOUT;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < NSUB; ++w) v += red[(w * CB + cc) * ABD_NOUT + k];  // waves w*CB + cc hold chain cc
    a.partials[((int64_t)(cbase + cc) * gridDim.x + blk) * ABD_NOUT + k] = v;  // rows in range order
  }
}
