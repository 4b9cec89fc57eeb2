// abd_resident.hpp -- the dense evaluation kernel kept RESIDENT on the device for one NUTS trajectory of one chain
// (included by abd_kernels.hpp after abd_dense.hpp).
//
// NUTS is sequential per chain: every leapfrog is one logp+grad evaluation at a new theta with the SAME discrete
// state (abd.py:922: one logp_dlogp per leapfrog), and what a chain sees is the latency of one evaluation call.
// Launched per evaluation, that call is launch (~5 us) + workgroup dispatch + per-range set-up that does not depend
// on theta at all (packed rows, constrain, 2^(j/1024) table: ~8 us) + gap loop + a second launch for the fixed-order
// sum (~6 us).  This kernel is launched ONCE per trajectory: every wave does the theta-independent set-up of its
// range once, keeps the constrained masks in LDS, and then serves evaluation COMMANDS that the host writes into a
// mailbox in mapped host memory:
//
//   host:  params of theta -> mailbox, then the command's sequence number (= the completion tag of its result row)
//   WG 0:  wave 0 polls the mailbox over PCIe, copies a new command into device memory (write-through stores) and
//          publishes its sequence number there
//   all:   wave 0 of every workgroup polls that device word, the workgroup rebuilds the power tables, walks its
//          ranges exactly as abd_dense_kernel<R, 1, true> does (same range table, same code, same order), writes its
//          partial row write-through and counts itself in with one device-scope atomic; the workgroup that counts
//          in last sums the partials in the fixed order of finalize_chain and writes row + tag to mapped host memory
//
// so one evaluation costs a PCIe poll, the tables, the gap loop and the hand-off -- and its 16 doubles are bit-identical
// to those of a launched evaluation with the same grid.  The command "sequence + 0.5" ends the kernel.
//
// Every wait has an exit: WG 0 gives up after cmd_timeout ticks of the 100 MHz s_memrealtime counter without a new
// command and tells the others (status word = expired; the host relaunches), the others give up on their own after
// guard_timeout.  The hand-off never waits.  The host launches at most as many resident workgroups as fit the chip
// together (abd_capi.hip), so no workgroup waits for one that cannot start.
#pragma once

#define ABD_RES_WORDS 16       // command: [0..6] [8..11] chain constants, [7] and [15] the sequence number (once per 64-byte line)
#define ABD_RES_SEQ_COPIES 32  // device copies of the published sequence number (pollers spread over them)
#define ABD_RES_SEQ_STRIDE 16  // 128 bytes apart: one memory channel each
#define ABD_RES_RELAY_WORDS (ABD_RES_WORDS + ABD_RES_SEQ_COPIES * ABD_RES_SEQ_STRIDE)
#define ABD_RES_PIECES 2       // pieces of a range whose masks are cached (the host checks the range table)

__host__ __device__ inline size_t abd_res_tables_bytes(int G) {
  const size_t tabs = (size_t)3 * (size_t)(G + 1) * sizeof(double2_t), fin = (size_t)ABD_FIN_PARTS * ABD_NOUT * sizeof(double);
  return tabs > fin ? tabs : fin;
}

struct ResidentArgs {
  EvalArgs a;                        // ch[0].rw / .waner: the chain's discrete state; range_tab: (grid.x, 4 ranges per workgroup); partials: [grid.x][ABD_NOUT]
  const unsigned long long* mail;    // mapped host memory, ABD_RES_WORDS words (128-byte aligned)
  unsigned long long* relay;         // device memory, ABD_RES_RELAY_WORDS words
  unsigned long long* done;          // device memory: partial rows written by all kernels of this unit so far
  unsigned long long done0;          // its value when this kernel starts
  double* out;                       // the result row (mapped host memory)
  unsigned int* status;              // mapped host memory: [0] = how the kernel ended (1 told to, 2 expired), [1] = commands served
  double seq0;                       // sequence numbers <= seq0 are stale
  unsigned long long cmd_timeout, guard_timeout;  // ticks of s_memrealtime
  int poll_naps;                     // pauses of ~0.25 us between two polls of the published sequence number
#ifdef ABD_STAMPS
  int dbg_launch, dbg_unit;
#endif
};

__device__ __forceinline__ double res_as_double(unsigned long long w) { return __longlong_as_double((long long)w); }
__device__ __forceinline__ unsigned long long res_as_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
// a wave-uniform double out of lane `l` of a 64-bit VGPR pair
__device__ __forceinline__ double res_readlane(unsigned long long w, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)w, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w >> 32), l);
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ double res_uniform(double v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ bool res_is_eval(double seq) { return seq == __builtin_floor(seq); }

#ifdef ABD_STAMPS
// diagnostic build (tools/probe_resident_phases.py): thread 0 of every workgroup adds up, over the commands it serves,
// the ticks of s_memrealtime it spends in each phase of a round and leaves the sums in EvalArgs::stamps[blockIdx.x][16]
#define ABD_RES_T(k)                                              \
  do {                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    ph[k] += now_ - tprev;                                        \
    tprev = now_;                                                 \
  } while (0)
#else
#define ABD_RES_T(k)
#endif

template <typename R>
__global__ __launch_bounds__(ABD_BLOCK, 4) void abd_dense_resident_kernel(const ResidentArgs ra) {
  const EvalArgs& a = ra.a;
  // LDS: [2][G+1] power tables, [G+1] ones table (together also the scratch of the fixed-order sum), block reduction,
  // 2^(j/1024) table, the command, the constrained masks of this workgroup's pieces
  extern __shared__ __align__(16) unsigned char smem[];
  const int G = a.G, N = a.N, nt = a.nt;
  const int tstride = G + 1;
  double2_t* tabs = reinterpret_cast<double2_t*>(smem);
  double2_t* tab_ones = tabs + 2 * tstride;
  // the tables' region is at least the ABD_FIN_PARTS x ABD_NOUT doubles the fixed-order sum needs (abd_res_tables_bytes)
  double* red = reinterpret_cast<double*>(smem + abd_res_tables_bytes(G));  // [WAVES][ABD_NOUT]
  double* tab_e2 = red + ABD_WAVES_PER_BLOCK * ABD_NOUT;                     // [ABD_EXP2_TAB]
  unsigned long long* cmd = reinterpret_cast<unsigned long long*>(tab_e2 + ABD_EXP2_TAB);  // [ABD_RES_WORDS]
  uint64_t* icache = reinterpret_cast<uint64_t*>(cmd + ABD_RES_WORDS);                       // [ABD_RES_PIECES][ABD_MAXT][ABD_BLOCK]
  int* flag = reinterpret_cast<int*>(icache + ABD_RES_PIECES * ABD_MAXT * ABD_BLOCK);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ChainPar& p = a.ch[0];

  const int nblk = (int)gridDim.x;
  const int xcd = (int)blockIdx.x % 8, q8 = nblk / 8, rem8 = nblk % 8;
  const int blk = a.xcd_remap ? xcd * q8 + min(xcd, rem8) + (int)blockIdx.x / 8 : (int)blockIdx.x;

  // ---- once: this wave's range, the packed rows of its (at most two) pieces, constrain (abd.py:640-667) ----
  const int4 rt = reinterpret_cast<const int4*>(a.range_tab)[blk * ABD_WAVES_PER_BLOCK + wave];
  const int lg0 = rt.x, g0_first = rt.y, rows = rt.z;
  const int rows_a = min(G - g0_first, rows);  // piece 0: lane group lg0, gaps [g0_first, g0_first + rows_a)
  const int rows_b = rows - rows_a;            // piece 1: lane group lg0 + 1, gaps [0, rows_b)
  bool wj0 = false, wj1 = false;
  int n1_0 = 0, n1_1 = 0;
#pragma unroll
  for (int pc = 0; pc < ABD_RES_PIECES; ++pc) {
    const int j = (lg0 + pc) * 64 + lane;
    uint64_t V[ABD_MAXT], P[ABD_MAXT], Rw[ABD_MAXT], I[ABD_MAXT];
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) V[t] = P[t] = Rw[t] = I[t] = 0;
    bool wj = false;
    int n1 = 0;
    if ((pc == 0 ? rows_a : rows_b) > 0 && j < N) {
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) {
        if (t < nt) {
          V[t] = a.vw[(int64_t)t * N + j];
          if (a.pw) P[t] = a.pw[(int64_t)t * N + j];
          Rw[t] = p.rw[(int64_t)t * N + j];
        }
      }
      wj = p.waner[j] != 0;
      constrain_masks(Rw, P, a, I);
#pragma unroll
      for (int t = 0; t < ABD_MAXT; ++t) n1 += __builtin_popcountll(Rw[t]);  // Bernoulli(i_raw | p) is on the RAW matrix
    }
#pragma unroll
    for (int t = 0; t < ABD_MAXT; ++t) icache[(pc * ABD_MAXT + t) * ABD_BLOCK + tid] = I[t];
    if (pc == 0) {
      wj0 = wj;
      n1_0 = n1;
    } else {
      wj1 = wj;
      n1_1 = n1;
    }
  }
  for (int e = tid; e < ABD_EXP2_TAB; e += ABD_BLOCK) tab_e2[e] = a.exp2_tab[e];

  const double2_t* tab_n = tabs;
  const double2_t* tab_sw = tabs + tstride;
  double last = ra.seq0;
  unsigned long long served = 0;

#ifdef ABD_STAMPS
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t_begin = tprev;
#endif
  for (;;) {
    // ---- wait for the next command (wave 0), leave it in LDS ----
    if (wave == 0) {
      unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long w = 0;
      double s = last;
      if (blockIdx.x == 0) {
        for (;;) {  // the mailbox, over PCIe: both 64-byte lines must carry the same new sequence number
          w = lane < ABD_RES_WORDS ? __hip_atomic_load(ra.mail + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
          const double sa = res_readlane(w, 7), sb = res_readlane(w, ABD_RES_WORDS - 1);
          if (sa == sb && sa > last) {
            s = sa;
            break;
          }
          if (__builtin_amdgcn_s_memrealtime() - t0 > ra.cmd_timeout) {
            s = __builtin_floor(last) + 0.25;  // not a whole number: everybody leaves
            if (lane == 0) __hip_atomic_store(ra.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
        // hand the command to the other workgroups: constants first (write-through, waited for), then the number
        if (lane < ABD_RES_WORDS - 1) __hip_atomic_store(ra.relay + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < ABD_RES_SEQ_COPIES)
          __hip_atomic_store(ra.relay + ABD_RES_WORDS + lane * ABD_RES_SEQ_STRIDE, res_as_bits(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const unsigned long long* my_seq = ra.relay + ABD_RES_WORDS + ((int)blockIdx.x % ABD_RES_SEQ_COPIES) * ABD_RES_SEQ_STRIDE;
        for (;;) {
          s = res_uniform(res_as_double(__hip_atomic_load(my_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
          if (s > last) break;
          if (__builtin_amdgcn_s_memrealtime() - t0 > ra.guard_timeout) {
            s = __builtin_floor(last) + 0.25;
            break;
          }
          for (int q = 0; q < ra.poll_naps; ++q) __builtin_amdgcn_s_sleep(8);
        }
        // the constants were complete in memory before the number was published, and this load is issued after the
        // number has come back
        if (res_is_eval(s))
          w = lane < ABD_RES_WORDS - 1 ? __hip_atomic_load(ra.relay + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
      if (lane < ABD_RES_WORDS) cmd[lane] = lane == ABD_RES_WORDS - 1 ? res_as_bits(s) : w;
    }
    __syncthreads();
    const double seq = res_uniform(res_as_double(cmd[ABD_RES_WORDS - 1]));
    if (!res_is_eval(seq)) break;  // told to leave (sequence + 0.5) or expired
    ABD_RES_T(0);

    // ---- the chain constants of this command; power tables: wave 0 rho_n, wave 1 rho_s, wave 2 the ones ----
    const double* cw = reinterpret_cast<const double*>(cmd);
    const double perm_n = res_uniform(cw[0]), temp_n = res_uniform(cw[1]), rho_n = res_uniform(cw[2]), init_n = res_uniform(cw[3]);
    const double perm_s = res_uniform(cw[4]), rho_s = res_uniform(cw[5]), init_s = res_uniform(cw[6]);
    const double b_n = res_uniform(cw[8]), d_n = res_uniform(cw[9]), b_s = res_uniform(cw[10]), d_s = res_uniform(cw[11]);
    if (wave == 0) fill_pow_table_wave(tabs, rho_n, tstride, lane);
    if (wave == 1) fill_pow_table_wave(tabs + tstride, rho_s, tstride, lane);
    if (wave == 2) fill_ones_table_wave(tab_ones, tstride, lane);
    const DenseChain kc = dense_chain(perm_n, temp_n, rho_n, init_n, perm_s, rho_s, init_s, b_n, d_n, b_s, d_s);
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    __syncthreads();
    ABD_RES_T(1);

    // ---- the pieces, exactly as abd_dense_kernel walks them ----
    {
      int lg = lg0, g0 = g0_first, rows_left = rows;
      for (int pc = 0; rows_left > 0; ++pc, ++lg, g0 = 0) {
        const int g1 = min(G, g0 + rows_left);
        rows_left -= g1 - g0;
        const int j = lg * 64 + lane;
        if (j < N) {
          // the vaccination words again from L2 (kept in registers across commands they would push the walk into scratch)
          uint64_t I[ABD_MAXT], V[ABD_MAXT];
#pragma unroll
          for (int t = 0; t < ABD_MAXT; ++t) {
            V[t] = t < nt ? a.vw[(int64_t)t * N + j] : 0ull;
            I[t] = icache[(pc * ABD_MAXT + t) * ABD_BLOCK + tid];
          }
          const bool wj = pc == 0 ? wj0 : wj1;
          if (g0 == 0) {  // each individual's gap 0 belongs to exactly one piece
            acc[ABD_NACC] += (double)(pc == 0 ? n1_0 : n1_1);
            acc[ABD_NACC + 1] += wj ? 1.0 : 0.0;
          }
          double tn = 0.0, dn = 0.0, ts = 0.0, ds = 0.0;
          uint32_t cfn_hi = 0, cfs_hi = 0;
          if (g0 > 0) dense_start_state(I, V, g0, tab_n, wj ? tab_sw : tab_ones, tn, dn, ts, ds, cfn_hi, cfs_hi);
          dense_walk<R, true>(a, kc, I, V, lg, lane, g0, g1, wj, tn, dn, ts, ds, cfn_hi, cfs_hi, tab_e2, acc);
        }
      }
    }

    // ---- lanes -> wave -> workgroup; the partial row goes out write-through; count in ----
#ifdef ABD_STAMPS
    if (acc[0] == 0x1.23456789abcdep+100) acc[1] += 1.0;  // keeps the stamp behind the walk
#endif
    ABD_RES_T(2);
    const double tot = wave_reduce16(acc, lane);
    if ((lane & 3) == 0) red[wave * ABD_NOUT + reduce16_index(lane)] = tot;
    __syncthreads();
    if (wave == 0) {
      if (lane < ABD_NOUT) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < ABD_WAVES_PER_BLOCK; ++w) v += red[w * ABD_NOUT + lane];
        __hip_atomic_store(a.partials + (int64_t)blk * ABD_NOUT + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row is in memory before this workgroup counts in
      if (lane == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(ra.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = old + 1ull == ra.done0 + (served + 1ull) * (unsigned long long)nblk ? 1 : 0;
      }
    }
    __syncthreads();
    ABD_RES_T(3);
    if (flag[0]) {
      // last one in: every row was in memory before its workgroup counted in, and the loads below are issued after
      // this workgroup's own count has come back
      finalize_chain_coherent<ABD_BLOCK>(a.partials, nblk, ra.out, reinterpret_cast<double*>(smem), tid, seq);
      if (tid == 0) __hip_atomic_store(ra.status + 1, (unsigned int)(served + 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();  // the scratch becomes the power tables again; flag and cmd are rewritten only behind this
#ifdef ABD_STAMPS
    ph[5] += flag[0] ? 1ull : 0ull;
    ph[6] += 1ull;
#endif
    ABD_RES_T(4);
    last = seq;
    served += 1ull;
  }
#ifdef ABD_STAMPS
  if (a.stamps && tid == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(a.stamps + (int64_t)blockIdx.x * 16 + k, ph[k]);  // summed over all kernels
  if (a.stamps && tid == 0 && blockIdx.x == 0) {  // log of launches: when this kernel ran
    unsigned long long* lg = a.stamps + 2048 * 16 + (int64_t)(ra.dbg_launch % 2048) * 4;
    lg[0] = t_begin;
    lg[1] = __builtin_amdgcn_s_memrealtime();
    lg[2] = (unsigned long long)ra.dbg_unit;
    lg[3] = ph[6];
  }
#endif
  if (blockIdx.x == 0 && tid == 0) {
    const double seq = res_as_double(cmd[ABD_RES_WORDS - 1]);
    if (seq - __builtin_floor(seq) == 0.5) __hip_atomic_store(ra.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
