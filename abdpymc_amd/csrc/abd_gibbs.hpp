// abd_gibbs.hpp -- one binary Gibbs-Metropolis sweep over [i_raw, ab_s_waner] on the device
// (included by abd_kernels.hpp after the shared helpers and the sparse kernel's `responses`).
//
// What it replaces: PyMC's BinaryGibbsMetropolis.astep on the two discrete variables of the model
// (reference abd.py:427, 373; step assignment made by pm.sample, abd.py:922): shuffle all G*N + N binary
// dims, and for each, with probability transit_p = 0.8, flip it, evaluate the JOINT logp and keep the flip
// with probability min(1, exp(delta)).  That is ~0.8 (G N + N) full-graph evaluations per draw.
//
// Why it factorises: flipping a bit of individual j changes (a) that individual's own likelihood terms and
// (b) a prior term whose change does not depend on any other bit: Bernoulli(i_raw | p) moves by
// +-(log p - log(1 - p)) = +-theta_0, Bernoulli(waner | p_waner) by +-theta_7.  So the sweeps of different
// individuals commute exactly, the relative order of one individual's dims under a uniform global shuffle is
// uniform, and the cross-individual order is irrelevant: one wave per (individual, chain) running that
// individual's G + 1 proposals in a uniformly random order IS the reference's sweep, at O(G) instead of
// O(G N) work per proposal.
//
// Randomness: Philox4x32-10, counter (dim, individual, chain slot id + offset, 0), key (seed_lo ^ sweep * 0x9E3779B9, seed_hi):
// word 0 orders the dims (low 9 bits replaced by the dim index, so keys are unique), word 1 is the transit
// draw, word 2 the acceptance draw.  oracle/abd_oracle.c restates the same stream, so whole trajectories can
// be compared bit for bit.
#pragma once

struct Philox4 {
  uint32_t w[4];
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                          uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.w[0] = c0;
  o.w[1] = c1;
  o.w[2] = c2;
  o.w[3] = c3;
  return o;
}

#define ABD_TRANSIT_P_U32 3435973836u  // floor(0.8 * 2^32): propose iff word 1 < this   (transit_p = 0.8)
#define ABD_GIBBS_WAVE_LDS 1856        // per wave: keys u32[260] + order u16[260] + transit u8[260], padded

struct GibbsArgs {
  EvalArgs e;  // panels, packed words, chain parameters (ch[k].rw / waner are updated IN PLACE)
  uint32_t seed_lo, seed_hi, sweep, pad_;
  uint32_t stream[ABD_MAX_BATCH_K];  // third counter word of each chain: its slot id (+ the caller's offset)
  double theta0[ABD_MAX_BATCH_K];  // log p - log(1 - p)             = p_logodds__
  double theta7[ABD_MAX_BATCH_K];  // log p_waner - log(1 - p_waner) = ab_s_p_waner_logodds__
  double is2_n[ABD_MAX_BATCH_K];   // 1 / sigma_n^2
  double is2_s[ABD_MAX_BATCH_K];
  unsigned long long* counts;      // [n_chains][2]: accepted, proposed (integer atomics: order-free)
};

__device__ __forceinline__ double readfirstlane_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// -1/2 sum (q / sigma)^2 over this individual's observations for the given masks (terms that do not depend
// on the discrete state are left out: they cancel in every difference)
template <typename R, bool DENSE>
__device__ __forceinline__ double individual_loglik(const EvalArgs& a, const ChainPar& p, int j, int lane,
                                                    const uint64_t I[ABD_MAXT], const uint64_t V[ABD_MAXT], bool wj,
                                                    const double2_t* tab_n, const double2_t* tab_sw,
                                                    const double2_t* tab_ones, double is2_n, double is2_s,
                                                    const YX<R> (&dn)[ABD_MAXT], const YX<R> (&ds)[ABD_MAXT]) {
  const double2_t* tab_s = wj ? tab_sw : tab_ones;
  double acc = 0.0;
  double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;  // unused gradient outputs
  if (DENSE) {
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) {
      if (t < a.nt) {
        const int g = t * 64 + lane;
        // padding lanes (g >= G) must stay inside the power tables: their residual is multiplied by 0,
        // but 0 * (garbage read past the table) could be NaN
        const Resp rs = responses(g < a.G ? g : a.G - 1, t + 1, I, V, tab_n, tab_s);
        const double guard = g < a.G ? 1.0 : 0.0;
        const double an = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
        const double as = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
        double q2n = 0.0, q2s = 0.0;
        obs_term<false>(an, (double)dn[t].x, (double)dn[t].y, p.b_n, p.d_n, guard, q2n, d0, d1, d2, d3);
        obs_term<false>(as, (double)ds[t].x, (double)ds[t].y, p.b_s, p.d_s, guard, q2s, d0, d1, d2, d3);
        acc = fma(-0.5 * is2_n, q2n, acc);
        acc = fma(-0.5 * is2_s, q2s, acc);
      }
    }
  } else {
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
        const Resp rs = responses(g, a.nt, I, V, tab_n, tab_s);
        const double guard = in ? 1.0 : 0.0;
        double q2 = 0.0;
        if (ag == 0) {
          const double an = p.init_n + (rs.cum_i ? p.perm_n : 0.0) + p.temp_n * rs.un;
          obs_term<false>(an, x, y, p.b_n, p.d_n, guard, q2, d0, d1, d2, d3);
          acc = fma(-0.5 * is2_n, q2, acc);
        } else {
          const double as = p.init_s + (rs.cum_iv ? p.perm_s : 0.0) + rs.us;
          obs_term<false>(as, x, y, p.b_s, p.d_s, guard, q2, d0, d1, d2, d3);
          acc = fma(-0.5 * is2_s, q2, acc);
        }
      }
    }
  }
  return readfirstlane_f64(wave_sum(acc));
}

template <typename R, bool DENSE>
__global__ __launch_bounds__(ABD_BLOCK) void abd_gibbs_kernel(const GibbsArgs ga) {
  // LDS: [2][G+1] power tables of the block's chain, [G+1] ones, then ABD_GIBBS_WAVE_LDS bytes per wave
  extern __shared__ __align__(16) unsigned char smem[];
  const EvalArgs& a = ga.e;
  const int G = a.G, N = a.N, nt = a.nt;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + 2 * tstride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* wbase = reinterpret_cast<unsigned char*>(tab_ones + tstride) + wave * ABD_GIBBS_WAVE_LDS;
  uint32_t* keyv = reinterpret_cast<uint32_t*>(wbase);           // [260] sort key by dim
  uint16_t* order = reinterpret_cast<uint16_t*>(wbase + 1040);   // [260] dim by rank
  unsigned char* transit = wbase + 1040 + 520;                   // [260] 1 = propose, by dim

  const int c = blockIdx.y;  // one chain per block row
  const ChainPar& p = a.ch[c];
  fill_pow_table(tabs, p.rho_n, tstride, tid, ABD_BLOCK);
  fill_pow_table(tabs + tstride, p.rho_s, tstride, tid, ABD_BLOCK);
  fill_ones_table(tab_ones, tstride, tid, ABD_BLOCK);
  __syncthreads();
  const double theta0 = ga.theta0[c], theta7 = ga.theta7[c], is2_n = ga.is2_n[c], is2_s = ga.is2_s[c];
  const uint32_t k0 = ga.seed_lo ^ (ga.sweep * 0x9E3779B9u), k1 = ga.seed_hi;
  const uint32_t cs = ga.stream[c];
  uint64_t* rw = const_cast<uint64_t*>(p.rw);
  int8_t* waner = const_cast<int8_t*>(p.waner);
  const int n_dims = G + 1;  // dims 0..G-1: i_raw[g, j]; dim G: ab_s_waner[j]
  unsigned long long n_acc = 0, n_prop = 0;

  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    // ---- this individual's discrete state and data ----
    uint64_t V[ABD_MAXT], P[ABD_MAXT], Rw[ABD_MAXT], I[ABD_MAXT];
    YX<R> dn[ABD_MAXT], ds[ABD_MAXT];
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) {
      V[t] = P[t] = Rw[t] = 0;
      dn[t].x = dn[t].y = ds[t].x = ds[t].y = 0;
      if (t < nt) {
        V[t] = uniform_word(a.vw, (int64_t)t * N + j);
        if (a.pw) P[t] = uniform_word(a.pw, (int64_t)t * N + j);
        Rw[t] = uniform_word(rw, (int64_t)t * N + j);
        if (DENSE) {
          const int g = min(t * 64 + lane, G - 1);
          dn[t] = reinterpret_cast<const YX<R>*>(a.yx_n)[(int64_t)g * N + j];  // one strided gather per sweep
          ds[t] = reinterpret_cast<const YX<R>*>(a.yx_s)[(int64_t)g * N + j];
        }
      }
    }
    bool wj = __builtin_amdgcn_readfirstlane((int)waner[j]) != 0;
    constrain_masks(Rw, P, a, I);

    // ---- random order and transit flags of this individual's dims ----
    for (int d = lane; d < n_dims; d += 64) {
      const Philox4 r = philox4x32_10((uint32_t)d, (uint32_t)j, cs, 0u, k0, k1);
      keyv[d] = (r.w[0] & ~0x1FFu) | (uint32_t)d;
      transit[d] = r.w[1] < ABD_TRANSIT_P_U32 ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < n_dims; d += 64) {  // rank = number of dims with a smaller key; order[rank] = dim
      const uint32_t mine = keyv[d];
      int rank = 0;
      for (int e = 0; e < n_dims; ++e) rank += keyv[e] < mine ? 1 : 0;
      order[rank] = (uint16_t)d;
    }
    __builtin_amdgcn_wave_barrier();

    double ll = individual_loglik<R, DENSE>(a, p, j, lane, I, V, wj, tabs, tabs + tstride, tab_ones, is2_n, is2_s, dn, ds);

    // ---- the sweep ----
    for (int k = 0; k < n_dims; ++k) {
      const int d = __builtin_amdgcn_readfirstlane((int)order[k]);
      if (!__builtin_amdgcn_readfirstlane((int)transit[d])) continue;  // same value proposed: nothing to do
      ++n_prop;
      double delta;
      uint64_t Rn[ABD_MAXT], In[ABD_MAXT];
      bool wn = wj;
      if (d < G) {
        const uint64_t bit = 1ull << (d & 63);
        bool was_one = false;
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t) {
          Rn[t] = Rw[t];
          if (t == (d >> 6)) {
            was_one = (Rw[t] & bit) != 0;
            Rn[t] ^= bit;
          }
        }
        delta = was_one ? -theta0 : theta0;  // Bernoulli(i_raw | p) on the RAW matrix (abd.py:427)
        constrain_masks(Rn, P, a, In);
      } else {
        wn = !wj;
        delta = wn ? theta7 : -theta7;  // Bernoulli(ab_s_waner | p_waner)   (abd.py:373)
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t) {
          Rn[t] = Rw[t];
          In[t] = I[t];
        }
      }
      bool same = wn == wj;
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) same = same && In[t] == I[t];
      double ll_new = ll;
      if (!same) {  // the constrained infections (or the waning class) changed: re-evaluate this individual
        ll_new = individual_loglik<R, DENSE>(a, p, j, lane, In, V, wn, tabs, tabs + tstride, tab_ones, is2_n, is2_s, dn, ds);
        delta += ll_new - ll;
      }
      // metrop_select: keep the flip if delta > 0 or delta > log(u)
      const Philox4 r = philox4x32_10((uint32_t)d, (uint32_t)j, cs, 0u, k0, k1);
      const double u = ((double)r.w[2] + 0.5) * (1.0 / 4294967296.0);
      if (delta > 0.0 || delta > log(u)) {
#pragma unroll
        for (int t = 0; t < ABD_MAXT; ++t) {
          Rw[t] = Rn[t];
          I[t] = In[t];
        }
        wj = wn;
        ll = ll_new;
        ++n_acc;
      }
    }

    // ---- write the individual's state back ----
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t)
        if (t < nt) rw[(int64_t)t * N + j] = Rw[t];
      waner[j] = wj ? 1 : 0;
    }
  }
  if (lane == 0 && (n_acc | n_prop)) {
    atomicAdd(ga.counts + 2 * c + 0, n_acc);
    atomicAdd(ga.counts + 2 * c + 1, n_prop);
  }
}

// packed words [nt][N] -> (G, N) gap-major int8, for reading a chain's i_raw back
__global__ __launch_bounds__(256) void abd_unpack_bits_kernel(const uint64_t* __restrict__ src, int8_t* __restrict__ dst,
                                                              int G, int N) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (j < N && g < G) dst[(int64_t)g * N + j] = (int8_t)((src[(int64_t)(g >> 6) * N + j] >> (g & 63)) & 1ull);
}
