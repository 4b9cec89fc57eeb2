// abd_gibbs.hpp -- one binary Gibbs-Metropolis sweep over [i_raw, ab_s_waner] on the device
// (included by abd_gibbs.hip after the shared helpers and the sparse kernel's `responses`).
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

#include "abd_device.hpp"

// This lane's share of -1/2 sum (q / sigma)^2 over the individual's observations for the given masks (terms
// that do not depend on the discrete state are left out: they cancel in every difference).
//
// Dense panels: lane = gap 64 t + lane of round t; out[t] = both antigens' terms of that gap, for rounds
// t >= r0 only -- the constraints and the responses are causal, so a flip whose first changed infection lies
// in round r0 leaves earlier rounds as they were.  The response at a gap is the recurrence
// T[g] = rho T[g-1] + e[g] unrolled per round: rho^(lane+1) x (T at the end of the previous round, a
// wave-uniform carry) + the exposures of THIS round's word at or before the lane (power table).
//
// PROPOSAL: the rounds of a proposed state are evaluated one at a time and the walk stops as soon as the
// proposal cannot be accepted any more.  Every term is <= 0, so the rounds not yet evaluated can raise the
// log-ratio by at most minus their CURRENT sums (suf: lane t = that bound for rounds >= t); once
// delta + bound < log u the outcome is settled.  A new infection moves the titers of its own round by a
// whole boost and is usually rejected there: ~1 round per proposal instead of ~3 at G = 200.  The test is
// exact (it never changes a decision), with a relative margin of 1e-9 against rounding in the bound.
template <typename R, bool PROPOSAL, int MT>
__device__ __forceinline__ void dense_rounds(const EvalArgs& a, const ChainPar& p, int lane, const uint64_t I[MT],
                                             const uint64_t V[MT], const double2_t* tab_n, const double2_t* tab_s,
                                             double pwn, double pws, double is2_n, double is2_s,
                                             const YX<R> (&dn)[MT], const YX<R> (&ds)[MT], int r0,
                                             double& cvn, double& cvs, double (&out)[MT],
                                             const double (&cur)[MT], double suf, double logu, double& delta, bool& dead) {
  // cvn / cvs: lane t holds the response carried INTO round t (lane 0: 0).  Kept as one vector register each
  // instead of 2 x 5 wave-uniform doubles: the masks already fill the scalar register file.
  double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;  // unused gradient outputs
  bool ci = false, civ = false;                   // any exposure in earlier rounds
  const uint64_t le = (2ull << lane) - 1ull;      // bits at or before this lane (lane 63: all ones)
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t < a.nt) {
      if (t >= r0 && !dead) {
        double un = pwn * readlane_f64(cvn, t), us = pws * readlane_f64(cvs, t);
        uint64_t m = I[t];
        while (m) {  // wave-uniform loop over this word's infections
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const int idx = min(max(lane - b + 1, 0), a.G);  // 0 = "in the future"; padding lanes stay in the table
          un += tab_n[idx].x;
          us += tab_s[idx].x;
        }
        m = V[t];
        while (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          const int idx = min(max(lane - b + 1, 0), a.G);
          us += tab_s[idx].x;
        }
        const bool cum_i = ci || (I[t] & le) != 0;
        const bool cum_iv = civ || ((I[t] | V[t]) & le) != 0;
        const double guard = t * 64 + lane < a.G ? 1.0 : 0.0;
        const double an = p.init_n + (cum_i ? p.perm_n : 0.0) + p.temp_n * un;
        const double as = p.init_s + (cum_iv ? p.perm_s : 0.0) + us;
        double q2n = 0.0, q2s = 0.0;
        obs_term<false>(an, (double)dn[t].x, (double)dn[t].y, p.b_n, p.d_n, guard, q2n, d0, d1, d2, d3);
        obs_term<false>(as, (double)ds[t].x, (double)ds[t].y, p.b_s, p.d_s, guard, q2s, d0, d1, d2, d3);
        out[t] = fma(-0.5 * is2_s, q2s, -0.5 * is2_n * q2n);
        if (t + 1 < MT) {
          cvn = lane == t + 1 ? readlane_f64(un, 63) : cvn;
          cvs = lane == t + 1 ? readlane_f64(us, 63) : cvs;
        }
        if (PROPOSAL) {
          delta += wave_sum_uniform(out[t] - cur[t]);
          const double rest = t + 1 < MT ? readlane_f64(suf, t + 1) : 0.0;
          if (delta + rest < logu - 1e-9 * (fabs(delta) + rest + 1.0)) dead = true;
        }
      }
      ci |= I[t] != 0;
      civ |= (I[t] | V[t]) != 0;
    }
  }
}

// Sparse lists: the individual's observations of BOTH antigens form one combined list (N first, then S), 64 per
// pass, one per lane.  The first pass -- the only one for the reference's cohorts (~12 + 12 observations per
// individual) -- is kept in registers for the whole sweep together with the lane's antigen-specific constants, so
// a proposal costs one response + one logistic term per lane and no memory traffic.
template <typename R>
struct ObsLane {
  int g;              // gap of the observation
  double y, x;        // od, log dilution
  double guard;       // 1 for a real observation, 0 for a padding lane
  bool is_s;          // S antigen (else N)
  double init, perm, temp, b, d, nh_is2;  // the antigen's constants: a = init + [exposed] perm + temp u ; -1/2 sigma^-2
};

template <typename R>
__device__ __forceinline__ ObsLane<R> load_obs_lane(const EvalArgs& a, const ChainPar& p, int j, int idx, double is2_n,
                                                    double is2_s) {
  ObsLane<R> o;
  const int kn0 = a.ptr_n[j], cnt_n = a.ptr_n[j + 1] - kn0;
  const int ks0 = a.ptr_s[j], cnt_s = a.ptr_s[j + 1] - ks0;
  o.is_s = idx >= cnt_n;
  const bool valid = idx < cnt_n + cnt_s;
  o.guard = valid ? 1.0 : 0.0;
  o.g = 0;
  o.y = o.x = 0.0;
  if (valid) {
    if (o.is_s) {
      const int k = ks0 + idx - cnt_n;
      o.g = a.g_s[k];
      o.y = ld<R>(a.y_s, k);
      o.x = ld<R>(a.x_s, k);
    } else {
      const int k = kn0 + idx;
      o.g = a.g_n[k];
      o.y = ld<R>(a.y_n, k);
      o.x = ld<R>(a.x_n, k);
    }
  }
  o.init = o.is_s ? p.init_s : p.init_n;
  o.perm = o.is_s ? p.perm_s : p.perm_n;
  o.temp = o.is_s ? 1.0 : p.temp_n;  // unit S boosts (Q1)
  o.b = o.is_s ? p.b_s : p.b_n;
  o.d = o.is_s ? p.d_s : p.d_n;
  o.nh_is2 = -0.5 * (o.is_s ? is2_s : is2_n);
  return o;
}

template <typename R, int MT>
__device__ __forceinline__ double obs_lane_term(const EvalArgs& a, const ObsLane<R>& o, const uint64_t I[MT],
                                                const uint64_t V[MT], const double2_t* tab_n, const double2_t* tab_s) {
  const double2_t* tb = o.is_s ? tab_s : tab_n;
  double u = 0.0;
  bool cum = false;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t < a.nt) {
      const int rel = o.g - t * 64;  // bits <= rel of word t are exposures at or before the observation's gap
      const uint64_t le = rel >= 63 ? ~0ull : (rel < 0 ? 0ull : ((2ull << rel) - 1ull));
      cum |= ((o.is_s ? (I[t] | V[t]) : I[t]) & le) != 0;
      uint64_t m = I[t];
      while (m) {  // wave-uniform loops over the set bits; table entry 0 is "in the future" = 0
        const int bpos = __builtin_ctzll(m);
        m &= m - 1;
        u += tb[max(rel - bpos + 1, 0)].x;
      }
      m = V[t];
      while (m) {
        const int bpos = __builtin_ctzll(m);
        m &= m - 1;
        const double v = tb[max(rel - bpos + 1, 0)].x;
        u += o.is_s ? v : 0.0;  // doses boost S only
      }
    }
  }
  const double resp = o.init + (cum ? o.perm : 0.0) + o.temp * u;
  double q2 = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
  obs_term<false>(resp, o.x, o.y, o.b, o.d, o.guard, q2, d0, d1, d2, d3);
  return o.nh_is2 * q2;
}

template <typename R, int MT>
__device__ __forceinline__ double sparse_terms(const EvalArgs& a, const ChainPar& p, int j, int lane, const uint64_t I[MT],
                                               const uint64_t V[MT], const double2_t* tab_n, const double2_t* tab_s,
                                               double is2_n, double is2_s, const ObsLane<R>& first, int n_obs) {
  double acc = obs_lane_term<R, MT>(a, first, I, V, tab_n, tab_s);
  for (int base = 64; base < n_obs; base += 64) {  // individuals with more than 64 observations: the rest from memory
    const ObsLane<R> o = load_obs_lane<R>(a, p, j, base + lane, is2_n, is2_s);
    acc += obs_lane_term<R, MT>(a, o, I, V, tab_n, tab_s);
  }
  return acc;
}

// a wave-uniform value the compiler cannot see through (nor hoist what is computed from it out of the loop it is made in)
__device__ __forceinline__ int gibbs_opaque_uniform(int x) {
  asm volatile("" : "+v"(x));
  return __builtin_amdgcn_readfirstlane(x);
}

template <typename R, bool DENSE, int MT>
__global__ __launch_bounds__(ABD_BLOCK) void abd_gibbs_kernel(const GibbsArgs ga) {
  // LDS: [2][G+1] power tables of the block's chain, [G+1] ones, then abd_gibbs_wave_lds(G) bytes per wave
  extern __shared__ __align__(16) unsigned char smem[];
  const EvalArgs& a = ga.e;
  const int G = a.G, N = a.N, nt0 = a.nt;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + 2 * tstride;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t nd = (size_t)G + 1;
  unsigned char* wbase = reinterpret_cast<unsigned char*>(tab_ones + tstride) + (size_t)wave * abd_gibbs_wave_lds(G);
  uint32_t* keyv = reinterpret_cast<uint32_t*>(wbase);                                    // [G+1] sort key by dim
  uint16_t* order = reinterpret_cast<uint16_t*>(wbase + abd_gibbs_pad16(4 * nd));         // [G+1] dim by rank
  unsigned char* transit = wbase + abd_gibbs_pad16(4 * nd) + abd_gibbs_pad16(2 * nd);     // [G+1] 1 = propose, by dim
  double* logu = reinterpret_cast<double*>(transit + abd_gibbs_pad16(nd));                // [G+1] log of the acceptance uniform, by dim

  const int c = blockIdx.y;  // one chain per block row
  const ChainPar& p = a.ch[c];
  fill_pow_table(tabs, p.rho_n, tstride, tid, ABD_BLOCK);
  fill_pow_table(tabs + tstride, p.rho_s, tstride, tid, ABD_BLOCK);
  fill_ones_table(tab_ones, tstride, tid, ABD_BLOCK);
  __syncthreads();
  const double theta0 = ga.theta0[c], theta7 = ga.theta7[c], is2_n = ga.is2_n[c], is2_s = ga.is2_s[c];
  // rho^(lane + 1) = table entry lane + 2 (only used when a previous round exists, i.e. G > 64 >= lane + 1)
  const double pwn = tabs[min(lane0 + 2, G)].x, pws = tabs[tstride + min(lane0 + 2, G)].x;
  const uint32_t k0_0 = ga.seed_lo ^ (ga.sweep * 0x9E3779B9u), k1_0 = ga.seed_hi;
  const uint32_t cs = ga.stream[c];
  uint64_t* rw = const_cast<uint64_t*>(p.rw);
  int8_t* waner = const_cast<int8_t*>(p.waner);
  uint64_t* iw = const_cast<uint64_t*>(p.iw);
  long long d_n1 = 0, d_m1 = 0;  // changes of sum(i_raw), sum(ab_s_waner) over this wave's individuals
  const int n_dims = G + 1;  // dims 0..G-1: i_raw[g, j]; dim G: ab_s_waner[j]
  unsigned long long n_acc = 0, n_prop = 0;

  const int waves_total = gridDim.x * ABD_WAVES_PER_BLOCK;
  for (int j = blockIdx.x * ABD_WAVES_PER_BLOCK + wave; j < N; j += waves_total) {
    // (wave-uniform, loop-invariant values that the compiler would otherwise derive masks and key schedules from in front of
    // the loop and keep, spilled, for the whole kernel are re-made opaque per individual: abd_gibbs2.hpp has the account)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int nt = gibbs_opaque_uniform(nt0);
    const uint32_t k0 = (uint32_t)gibbs_opaque_uniform((int)k0_0), k1 = (uint32_t)gibbs_opaque_uniform((int)k1_0);
    // ---- this individual's discrete state and data ----
    uint64_t V[MT], P[MT], Rw[MT], I[MT];
    YX<R> dn[MT], ds[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      V[t] = P[t] = Rw[t] = 0;
      dn[t].x = dn[t].y = ds[t].x = ds[t].y = 0;
      if (t < nt) {
        V[t] = uniform_word(a.vw, (int64_t)t * N + j);
        if (a.pw) P[t] = uniform_word(a.pw, (int64_t)t * N + j);
        Rw[t] = uniform_word(rw, (int64_t)t * N + j);
        if (DENSE) {
          const int g = min(t * 64 + lane, G - 1);
          dn[t] = reinterpret_cast<const YX<R>*>(a.yxi_n)[(int64_t)j * G + g];  // the individual's gap axis: contiguous
          ds[t] = reinterpret_cast<const YX<R>*>(a.yxi_s)[(int64_t)j * G + g];
        }
      }
    }
    bool wj = __builtin_amdgcn_readfirstlane((int)waner[j]) != 0;
    constrain_masks<MT>(Rw, P, a, I);
    int pc0 = wj ? (1 << 16) : 0;  // sum(i_raw) and ab_s_waner of this individual before the sweep
#pragma unroll
    for (int t = 0; t < MT; ++t) pc0 += __builtin_popcountll(Rw[t]);

    // ---- random order and transit flags of this individual's dims ----
    for (int d = lane; d < n_dims; d += 64) {
      const Philox4 r = philox4x32_10((uint32_t)d, (uint32_t)j + ga.ind_offset, cs, 0u, k0, k1);
      keyv[d] = (r.w[0] & ~0x1FFu) | (uint32_t)d;
      transit[d] = r.w[1] < ABD_TRANSIT_P_U32 ? 1 : 0;
      logu[d] = log(((double)r.w[2] + 0.5) * (1.0 / 4294967296.0));  // one log per lane and dim, not one per proposal
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < n_dims; d += 64) {  // rank = number of dims with a smaller key; order[rank] = dim
      const uint32_t mine = keyv[d];
      int rank = 0;
      for (int e = 0; e < n_dims; ++e) rank += keyv[e] < mine ? 1 : 0;
      order[rank] = (uint16_t)d;
    }
    __builtin_amdgcn_wave_barrier();

    // sparse lists: this lane's observation of the first pass and the individual's observation count
    ObsLane<R> first;
    int n_obs = 0;
    if (!DENSE) {
      first = load_obs_lane<R>(a, p, j, lane, is2_n, is2_s);
      n_obs = __builtin_amdgcn_readfirstlane((a.ptr_n[j + 1] - a.ptr_n[j]) + (a.ptr_s[j + 1] - a.ptr_s[j]));
    }

    // this lane's terms at the current state (dense: by round of 64 gaps, with the responses carried into
    // each round; sparse: one sum in cur[0])
    double cur[MT], cur_cn = 0.0, cur_cs = 0.0;
#pragma unroll
    for (int t = 0; t < MT; ++t) cur[t] = 0.0;
    double suf = 0.0;  // lane t: -(sum of the current terms of rounds >= t) >= 0, the most those rounds can give back
    auto refresh_bounds = [&]() {
      double accb = 0.0;
      suf = 0.0;
#pragma unroll
      for (int t = MT - 1; t >= 0; --t) {
        if (t < nt) accb -= wave_sum_uniform(cur[t]);
        suf = lane == t ? accb : suf;
      }
    };
    if (DENSE) {
      double unused_delta = 0.0;
      bool unused_dead = false;
      dense_rounds<R, false, MT>(a, p, lane, I, V, tabs, wj ? tabs + tstride : tab_ones, pwn, wj ? pws : 1.0, is2_n, is2_s, dn, ds,
                             0, cur_cn, cur_cs, cur, cur, 0.0, 0.0, unused_delta, unused_dead);
      refresh_bounds();
    } else {
      cur[0] = sparse_terms<R, MT>(a, p, j, lane, I, V, tabs, wj ? tabs + tstride : tab_ones, is2_n, is2_s, first, n_obs);
    }

    // ---- the sweep ----
    for (int k = 0; k < n_dims; ++k) {
      const int d = __builtin_amdgcn_readfirstlane((int)order[k]);
      if (!__builtin_amdgcn_readfirstlane((int)transit[d])) continue;  // same value proposed: nothing to do
      ++n_prop;
      double delta;
      uint64_t In[MT];
      bool wn = wj;
      const uint64_t bit = d < G ? 1ull << (d & 63) : 0ull;  // the proposed flip of i_raw, in word d >> 6
      if (d < G) {
        uint64_t Rn[MT];
        bool was_one = false;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          Rn[t] = Rw[t];
          if (t == (d >> 6)) {
            was_one = (Rw[t] & bit) != 0;
            Rn[t] ^= bit;
          }
        }
        delta = was_one ? -theta0 : theta0;  // Bernoulli(i_raw | p) on the RAW matrix (abd.py:427)
        constrain_masks<MT>(Rn, P, a, In);
      } else {
        wn = !wj;
        delta = wn ? theta7 : -theta7;  // Bernoulli(ab_s_waner | p_waner)   (abd.py:373)
#pragma unroll
        for (int t = 0; t < MT; ++t) In[t] = I[t];
      }
      // first round of 64 gaps whose constrained infections differ (a waning flip touches every round)
      int r0 = wn == wj ? MT : 0;
#pragma unroll
      for (int t = MT - 1; t >= 0; --t)
        if (In[t] != I[t]) r0 = min(r0, t);
      double nxt[MT], nxt_cn = cur_cn, nxt_cs = cur_cs;
#pragma unroll
      for (int t = 0; t < MT; ++t) nxt[t] = cur[t];
      const double log_u = readfirstlane_f64(logu[d]);
      bool dead = false;  // dense: settled as a rejection before all rounds were evaluated
      if (r0 < MT) {  // something changed: re-evaluate this individual from there on
        if (DENSE) {
          dense_rounds<R, true, MT>(a, p, lane, In, V, tabs, wn ? tabs + tstride : tab_ones, pwn, wn ? pws : 1.0, is2_n, is2_s, dn,
                                ds, r0, nxt_cn, nxt_cs, nxt, cur, suf, log_u, delta, dead);
        } else {
          nxt[0] = sparse_terms<R, MT>(a, p, j, lane, In, V, tabs, wn ? tabs + tstride : tab_ones, is2_n, is2_s, first, n_obs);
          delta += wave_sum_uniform(nxt[0] - cur[0]);
        }
      }
      // metrop_select: keep the flip if delta > 0 or delta > log(u)
      if (!dead && (delta > 0.0 || delta > log_u)) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if (t == (d >> 6)) Rw[t] ^= bit;
          I[t] = In[t];
          cur[t] = nxt[t];
        }
        cur_cn = nxt_cn;
        cur_cs = nxt_cs;
        if (DENSE) refresh_bounds();
        wj = wn;
        ++n_acc;
      }
    }

    // ---- write the individual's state back: raw bits, waning flag, and what the slot keeps beside them (the
    // constrained words the evaluation kernels read, the changes of sum(i_raw) and sum(ab_s_waner)) ----
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if (t < nt) {
          rw[(int64_t)t * N + j] = Rw[t];
          iw[(int64_t)t * N + j] = I[t];
        }
      waner[j] = wj ? 1 : 0;
    }
    int pc1 = wj ? (1 << 16) : 0;
#pragma unroll
    for (int t = 0; t < MT; ++t) pc1 += __builtin_popcountll(Rw[t]);
    d_n1 += (pc1 & 0xFFFF) - (pc0 & 0xFFFF);
    d_m1 += (pc1 >> 16) - (pc0 >> 16);
  }
  if (lane0 == 0 && (n_acc | n_prop)) {
    atomicAdd(ga.counts + 2 * c + 0, n_acc);
    atomicAdd(ga.counts + 2 * c + 1, n_prop);
  }
  if (lane0 == 0 && (d_n1 | d_m1)) {
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(const_cast<long long*>(p.cnt));
    atomicAdd(cnt + 0, (unsigned long long)d_n1);  // two's complement: a negative change wraps to the right sum
    atomicAdd(cnt + 1, (unsigned long long)d_m1);
  }
}

