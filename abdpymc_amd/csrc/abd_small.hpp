// abd_small.hpp -- the small kernels of the context: packing / unpacking / flipping the discrete state, the recorded
// Deterministics (abd.py:649/667, 341, 389-391), the hardware-queue probe.
#pragma once

#include "abd_device.hpp"

// One wave that stays on the device for `ticks` of the 100 MHz s_memrealtime counter and says when (abd_context.hip:
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

// What the integer pre-pass needs besides the words (a 2 KB EvalArgs is not passed for it)
struct ConstrainArgs {
  const uint64_t* pw;  // [nt][N] packed pcrpos, nullptr = ignore_pcrpos
  int32_t N, nt, n_chunks, pad_;
  uint64_t chunk_mask[3][ABD_MAXT_MAX];
};

// Refresh a chain slot's cached state from its raw discrete state: iw = constrain(rw, pcrpos) (abd.py:640-667) for every
// individual, cnt[0] += sum(i_raw), cnt[1] += sum(ab_s_waner) (the caller zeroes cnt; integer adds: order-free).
template <int MT>
__global__ __launch_bounds__(256) void abd_constrain_kernel(const ConstrainArgs a, const uint64_t* __restrict__ rw,
                                                            const int8_t* __restrict__ waner, uint64_t* __restrict__ iw,
                                                            unsigned long long* cnt) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int n1 = 0, m1 = 0;
  if (j < a.N) {
    uint64_t P[MT], Rw[MT], I[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      P[t] = Rw[t] = 0;
      if (t < a.nt) {
        Rw[t] = rw[(int64_t)t * a.N + j];
        if (a.pw) P[t] = a.pw[(int64_t)t * a.N + j];
        n1 += __builtin_popcountll(Rw[t]);
      }
    }
    constrain_masks<MT>(Rw, P, a, I);
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < a.nt) iw[(int64_t)t * a.N + j] = I[t];
    m1 = waner[j] != 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n1 += __shfl_xor(n1, off, 64);
    m1 += __shfl_xor(m1, off, 64);
  }
  if ((threadIdx.x & 63) == 0 && (n1 | m1)) {
    atomicAdd(cnt + 0, (unsigned long long)n1);
    atomicAdd(cnt + 1, (unsigned long long)m1);
  }
}

// One proposed flip of the resident state (abd_flip_discrete): the bit, its individual's constrained words, the counters
template <int MT>
__global__ void abd_flip_kernel(const ConstrainArgs a, uint64_t* rw, int8_t* waner, uint64_t* iw, unsigned long long* cnt, int G,
                                int64_t flat) {
  const int N = a.N;
  const int64_t gn = (int64_t)G * N;
  if (flat < gn) {
    const int64_t g = flat / N, j = flat % N;
    const uint64_t bit = 1ull << (g & 63);
    const uint64_t w = rw[(g >> 6) * N + j] ^ bit;
    rw[(g >> 6) * N + j] = w;
    cnt[0] += (w & bit) ? 1ull : ~0ull;  // +1 / -1
    uint64_t P[MT], Rw[MT], I[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      P[t] = Rw[t] = 0;
      if (t < a.nt) {
        Rw[t] = rw[(int64_t)t * N + j];
        if (a.pw) P[t] = a.pw[(int64_t)t * N + j];
      }
    }
    constrain_masks<MT>(Rw, P, a, I);
#pragma unroll
    for (int t = 0; t < MT; ++t)
      if (t < a.nt) iw[(int64_t)t * N + j] = I[t];
  } else {
    const int8_t w = waner[flat - gn] ^ 1;
    waner[flat - gn] = w;
    cnt[1] += w ? 1ull : ~0ull;
  }
}

// Deterministics "i", "ab_n_mu", "ab_s_mu" for one chain, written (G, N) gap-major as PyMC records them;
// with `sums` ([3][G*N]: i, ab_n_mu, ab_s_mu) they are also added to running sums (posterior means on device).
template <int MT>
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
    uint64_t V[MT], I[MT];  // vaccinations; the chain's constrained infections (kept with the slot)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      V[t] = I[t] = 0;
      if (t < nt) {
        V[t] = uniform_word(a.vw, (int64_t)t * N + j);
        I[t] = uniform_word(p.iw, (int64_t)t * N + j);
      }
    }
    const bool wj = __builtin_amdgcn_readfirstlane((int)p.waner[j]) != 0;
    const double2_t* ts = wj ? tabs + tstride : tab_ones;
    for (int t = 0; t < nt; ++t) {
      const int g = t * 64 + lane;
      if (g < G) {
        const Resp rs = responses<MT>(g, t + 1, I, V, tabs, ts);
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


// packed words [nt][N] -> (G, N) gap-major int8, for reading a chain's i_raw back
__global__ __launch_bounds__(256) void abd_unpack_bits_kernel(const uint64_t* __restrict__ src, int8_t* __restrict__ dst,
                                                              int G, int N) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (j < N && g < G) dst[(int64_t)g * N + j] = (int8_t)((src[(int64_t)(g >> 6) * N + j] >> (g & 63)) & 1ull);
}

