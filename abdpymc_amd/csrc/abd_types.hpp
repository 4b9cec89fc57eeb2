// abd_types.hpp -- constants and plain structs shared by the device code and its host launchers.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else  // plain C++ (tests/native compiles abd_terms.hpp with g++): the qualifiers mean nothing there
#define __host__
#define __device__
#define __forceinline__ inline
#endif
#include <stddef.h>
#include <stdint.h>

#define ABD_MAXT 4          // 64-gap words per individual the register-resident kernels are built for by default (G <= 256) ...
#define ABD_MAXT_MAX 8      // ... and at most (G <= 512): the kernels that hold an individual's words in registers are
                            // templates over the word count (4 or 8); the dense evaluation kernel reads words on demand
#define ABD_NACC 13         // floating sums per chain (see enum below)
#define ABD_NOUT 16         // ABD_NACC + n1 + m1, padded
#define ABD_MAX_BATCH_K 16  // chains per launch
#define ABD_WAVES_PER_BLOCK 4
#define ABD_BLOCK (64 * ABD_WAVES_PER_BLOCK)

// raw sums accumulated on the device; per-chain constants are applied on the host (abd_terms.hpp: assemble_terms)
//   q = od - d s,  s = 1 / (1 + exp(b (a - x))),  h' = q s (1 - s)
//   d ll / d a = -b (d / sigma^2) h'
enum {
  A_N_Q2 = 0,   // sum q^2                        -> ll, d/d log sigma
  A_N_H,        // sum h'            -> init_n
  A_N_HC,       // sum h' [cumI>0]   -> perm_n
  A_N_HU,       // sum h' U_n        -> temp_n    U_n = sum_r rho_n^(g-r)
  A_N_HD,       // sum h' dU_n/drho  -> rho_n
  A_N_HX,       // sum h' (a - x)    -> b_n
  A_N_QS,       // sum q s           -> d_n
  A_S_Q2,
  A_S_H,
  A_S_HC,
  A_S_HD,
  A_S_HX,
  A_S_QS,
};

struct ChainPar {
  double perm_n, temp_n, rho_n, init_n;
  double perm_s, rho_s, init_s;
  double b_n, d_n;
  double b_s, d_s;
  const uint64_t* rw;   // [nt][N] packed i_raw of the chain
  const int8_t* waner;  // [N]
  // kept with the slot's discrete state and refreshed by whoever rewrites it (abd_small.hpp: abd_constrain_kernel,
  // abd_flip_kernel; the sweep kernels): the CONSTRAINED infections i = constrain(i_raw, pcrpos) (abd.py:640-667), packed
  // like rw, and {sum(i_raw), sum(ab_s_waner)}
  const uint64_t* iw;
  const long long* cnt;
};

// ---- leapfrog trains (abd_dense.hpp: train_epilogue; abd_sampler.hip) ----
// Inside one half of a NUTS tree doubling the leapfrogs follow each other deterministically (abd_nuts.hpp: stage_leapfrog /
// feed), so the launch that evaluates point k can itself assemble logp and gradient, finish leapfrog k, take the drift of
// leapfrog k + 1 and leave point k + 1 where the NEXT launch -- already queued behind it on the same stream -- finds it:
// the host round trip (result over PCIe, state machine, launch, dispatch: ~12 us) leaves the chain's critical path.
#define ABD_NT 17  // = ABD_N_THETA (abd_hip.h)
struct TrainPoint {         // a point of a train as the launch that evaluates it needs it
  double theta[ABD_NT];
  double p_half[ABD_NT];    // momentum after the first half kick of the leapfrog that leads to theta
  double tr[ABD_NT];        // backward transforms of theta (abd_terms.hpp: Transformed, indexed like theta)
  double L0[4], L1[4];      // -softplus(-t), -softplus(t) of theta[0], [3], [6], [7]
  // what the launch that left this point found at ITS point: the launch that evaluates this one passes it on to the host
  // (TrainArgs::fwd_rec), so that no launch of a train waits for its own PCIe writes before the next one may start
  double prev_lp, prev_g[ABD_NT];
};
struct TrainRecord {        // what the host takes from a launch (mapped host memory)
  double lp;
  double g[ABD_NT];
  double next_theta[ABD_NT], next_p_half[ABD_NT];  // the point the launch left for its successor
  double tag;               // completion tag, written last behind a system-scope fence
};
struct TrainArgs {
  TrainPoint* slots;        // device memory, [2]: written by one launch of the train, read by the next
  TrainRecord* rec;         // this launch's record
  double tag;
  double ve;                // direction x step size of the half
  double prior_const;       // theta-independent part of the priors (abd_context.hip: prior_constant)
  int32_t enabled;          // 0: an ordinary evaluation
  int32_t use_slot;         // >= 0: this launch's point is slots[use_slot]; < 0: `first` below (the host staged it)
  int32_t next_slot;        // where this launch leaves the next point
  int32_t own_record;       // 1: this launch writes its own record to `rec` (no successor is going to pass it on)
  int32_t dense;            // 1: the sums come from the dense kernel (its sum for d/db is scaled: abd_terms.hpp), 0: observation lists
  int32_t pad_;
  TrainRecord* fwd_rec;     // use_slot >= 0: the predecessor's record, written by this launch from slots[use_slot] ...
  double fwd_tag;           // ... under the predecessor's tag; nullptr: the predecessor wrote its own
  double inv_mass[ABD_NT];  // diagonal of M^-1
  TrainPoint first;
};

// ---- leapfrog trains of dense cohorts (abd_train.hpp; abd_sampler.hip) ----
// A train UNIT is 1, 2 or 4 chains that share their launches: one launch takes every stepping chain of the unit one
// leapfrog further (the chains read the same panel rows, once).  What persists between launches lives in device memory,
// one TrainChain per chain, and the launch's last workgroup runs the chain's state machine on it: assemble logp and
// gradient, finish the leapfrog, and -- the directions of all doublings of a transition are drawn when it begins
// (abd_nuts.hpp: dir_bits) -- go on into the next half of the tree when one is complete.  The host's tree logic follows
// behind on the records and stops asking for steps when the tree has ended; a new transition is handed over in a TrainBegin
// block (mapped host memory), read by the last workgroup of a launch in which the chain does not step.
#define ABD_TRAIN_CB 4      // chains per unit at most (= waves of a workgroup: wave k of the last workgroup runs chain k's state machine)
#define ABD_TRAIN_RING 64   // records per chain in mapped host memory
#define ABD_TRAIN_SHARDS 32      // a train launch's workgroups count in in shards (abd_dense.hpp: dense_body) ...
#define ABD_TRAIN_ONE_LEVEL 256  // ... unless the launch has at most this many: then one counter, and its last workgroup sums every row
#define ABD_TRAIN_CNT_STRIDE 32  // ... whose counters lie 128 bytes apart: [0] the top, [(1 + s) * stride] shard s
enum { ABD_TR_SKIP = 0, ABD_TR_STEP = 1, ABD_TR_BEGIN = 2 };  // what a launch does with a chain of its unit
enum { ABD_PH_IDLE = 0, ABD_PH_EVAL0 = 1, ABD_PH_LEAF = 2 };  // TrainChain::phase
struct TrainEnd {            // one end of the trajectory: point, momentum, gradient (abd_nuts.hpp: Phase)
  double q[ABD_NT], p[ABD_NT], g[ABD_NT];
};
struct TrainBegin {          // a new transition, staged by the host (abd_nuts.hpp: begin_draw / begin_finish)
  double q0[ABD_NT], p0[ABD_NT], g0[ABD_NT];  // g0 is not read when eval_first
  double inv_mass[ABD_NT];
  double eps;
  uint32_t dirs;             // bit d: the doubling at depth d goes forward
  int32_t max_depth;         // 0: no leapfrog at all (eval_first: just the evaluation at q0)
  int32_t eval_first;        // 1: the discrete state changed under the chain (abd.py:922: the sweep): evaluate at q0 first,
  int32_t pad_;              //    record that, then start the tree from (q0, p0, gradient found)
};
struct TrainChain {
  TrainPoint pt[2];          // pt[s]: the point a step launch with use_slot = s evaluates; written by the launch before it
  TrainEnd end[2];           // [0] the backward end of the trajectory, [1] the forward end
  double inv_mass[ABD_NT];
  double eps;
  uint32_t dirs;
  int32_t max_depth;
  int32_t phase, depth, n_leaf, n_target, dir;  // dir: +1 / -1, direction of the half being built
  int32_t pad_;
};
struct TrainChainArgs {      // one chain of a train launch
  TrainChain* st;
  TrainRecord* ring;         // mapped host memory, [ABD_TRAIN_RING]: record k at ring[k % ABD_TRAIN_RING] under tag k + 1
  const TrainBegin* begin;   // mapped host memory (action BEGIN)
  const uint64_t* iw;        // the chain slot's discrete state (ChainPar)
  const long long* cnt;
  const int8_t* waner;
  int32_t action;            // ABD_TR_*
  int32_t use_slot;          // STEP: the point is st->pt[use_slot]; the next one goes to pt[use_slot ^ 1]
  int32_t own;               // STEP: the launch writes the step's record itself (nothing queued behind it would pass it on)
  int32_t fwd_slot;          // >= 0: the chain's previous step left its record beside pt[fwd_slot]: the service workgroup passes it on
  int64_t rec_idx;           // STEP: index of the record this step produces
  int64_t fwd_idx;           // fwd_slot >= 0: index of the record passed on
};
struct DenseTrainArgs {
  const void* yx_n;          // panels, as in EvalArgs
  const void* yx_s;
  const void* od_n;
  const void* od_s;
  const uint8_t* xc_n;
  const uint8_t* xc_s;
  const double* dict_n;
  const double* dict_s;
  const uint64_t* vw;
  const double* exp2_tab;
  double* partials;          // [CB][workgroups][ABD_NOUT]
  unsigned int* fin_count;   // one zeroed counter: the workgroup that counts in last runs the state machines
#ifdef ABD_STAMPS
  unsigned long long* stamps;
#endif
  double prior_const;
  int32_t n_dict_n, n_dict_s;
  int32_t rg_base, rg_extra, rg_e_fin, rg_n_short;
  uint32_t rg_g_magic;
  int32_t xcd_remap;
  int32_t service;           // 1: workgroup 0 has no range: it passes records on to the host (TrainChainArgs::fwd_slot)
  int32_t G, N, n_lg, K_n, K_s;
  TrainChainArgs tc[ABD_TRAIN_CB];
};

struct EvalArgs {
  // sparse observation lists (CSR by individual)
  const void* y_n;
  const void* x_n;
  const void* y_s;
  const void* x_s;
  const uint16_t* g_n;  // gap of each observation
  const uint16_t* g_s;
  const int32_t* ptr_n;
  const int32_t* ptr_s;
  const int32_t* j_n;  // individual of each observation (observation-lane kernel)
  const int32_t* j_s;
  int32_t K_n, K_s;          // list lengths
  int32_t ob_n, ob_s, ob_c;  // observation-lane kernel: workgroups over the N list, the S list, the individuals
  int32_t pad3_;
  // dense panels, gap-major [G][N] of {od, log_dilution}
  const void* yx_n;
  const void* yx_s;
  // ... and individual-major [N][G] copies for the sweep kernels, which read ONE individual's whole gap axis at a time (3.2 KB
  // contiguous per antigen at 200 gaps, fp64; from the gap-major panel the same read touches 200 cache lines)
  const void* yxi_n;
  const void* yxi_s;
  // the same panels split for launches that evaluate one chain (abd_dense.hpp: XC): od [lane group][G][64] in the storage type, a
  // one-byte code per cell into the antigen's dictionary of distinct log dilutions (nullptr: more than 256 distinct values)
  const void* od_n;
  const void* od_s;
  const uint8_t* xc_n;
  const uint8_t* xc_s;
  const double* dict_n;
  const double* dict_s;
  int32_t n_dict_n, n_dict_s;
  // packed indicator panels [nt][N]
  const uint64_t* vw;
  const uint64_t* pw;  // nullptr = ignore_pcrpos
#ifdef ABD_STAMPS
  unsigned long long* stamps;  // diagnostic build: [grid.x][16] s_memrealtime at phase boundaries
#endif
  // dense kernel: how this launch shape cuts the (lane group, gap) plane into ranges (abd_eval.hip: range_split): range r
  // starts at row r base + min(r, extra) - e_fin min(r, n_short) of the flattened plane; row / G by g_magic = ceil(2^32 / G)
  // (exact below 2^32 / G rows: abd_create checks).  Closed form, not a table: a table is one more dependent memory
  // round trip in front of everything else a workgroup loads
  int32_t rg_base, rg_extra, rg_e_fin, rg_n_short;
  uint32_t rg_g_magic;
  const double* exp2_tab;  // dense kernel: 2^(j/1024), j = 0..1023, correctly rounded (copied to LDS per workgroup)
  double* partials;    // [n_chains][grid.x][ABD_NOUT]
  // dense kernel only: the fixed-order sum of the PREVIOUS launch's partials, done by the first
  // prev_n_chains workgroups of this launch (saves a kernel and a boundary per step when launches are
  // stream-ordered); prev_n_chains = 0 -> nothing to do
  const double* prev_partials;
  double* prev_out;
  int32_t prev_n_chains, prev_blocks;
  int32_t fin_rows;    // gap rows the finalizing workgroups are excused from
  int32_t xcd_remap;   // dense kernel: workgroup -> range mapping that keeps neighbouring ranges on one XCD
  double prev_tag;          // completion tag of the previous launch (see finalize_chain)
  // observation-lane kernel: the workgroup of a chain that finishes last sums that chain's partial rows itself
  // (abd_obs.hpp) -- one launch per evaluation instead of two.  fin_count: one zeroed counter per grid row.
  unsigned int* fin_count;
  unsigned int* fin_count2;  // dense kernel, a grid of more than ABD_TRAIN_ONE_LEVEL workgroups: counters of the two-level count-in, one set per grid row (abd_dense.hpp: two_level_sums)
  double* fin_out;
  double fin_tag;
  int32_t G, N, nt, n_chunks;
  int32_t n_chains, n_lg;     // n_lg: 64-individual lane groups (dense kernel)
  uint64_t chunk_mask[3][ABD_MAXT_MAX];
  ChainPar ch[ABD_MAX_BATCH_K];
  TrainArgs train;  // dense kernel, one chain per launch (abd_sampler.hip)
};

// row / G of the dense kernel's range arithmetic (abd_dense.hpp: range_of) by a 32-bit reciprocal: magic = ceil(2^32 / G),
// row / G = (row * magic) >> 32.  With e = magic G - 2^32 (0 <= e < G) the quotient is exact as long as row e < 2^32;
// abd_create keeps a cohort on the dense path only if that holds for every row of its (lane group, gap) plane, n_rows
// included (the end of the last range) -- abd_div_magic_exact; tests/native/magic_harness.cpp sweeps the boundary.
__host__ __device__ inline uint32_t abd_div_magic(uint32_t G) { return G > 1 ? (uint32_t)((((uint64_t)1 << 32) + G - 1) / G) : 0u; }
__host__ __device__ inline uint32_t abd_div_by_magic(uint32_t row, uint32_t magic) { return (uint32_t)(((uint64_t)row * magic) >> 32); }
__host__ __device__ inline bool abd_div_magic_exact(uint64_t n_rows, uint32_t G) {
  if (G <= 1) return n_rows < ((uint64_t)1 << 31);
  const uint64_t e = (uint64_t)abd_div_magic(G) * G - ((uint64_t)1 << 32);
  return n_rows < ((uint64_t)1 << 31) && n_rows * e < ((uint64_t)1 << 32);
}

struct double2_t {
  double x, y;
};

template <typename R>
struct YX {
  R y, x;
};


#define ABD_EXP2_TAB 1024  // entries of the 2^(j/1024) table (8 KB of LDS per workgroup)
#define ABD_XDICT 256      // distinct log dilutions per antigen the split panels can code (one byte per cell; abd_dense.hpp: XC)

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

#ifndef ABD_TRANSIT_P_U32  // (a diagnostic build may set it to 0: a sweep that proposes nothing measures the per-individual set-up)
#define ABD_TRANSIT_P_U32 3435973836u  // floor(0.8 * 2^32): propose iff word 1 < this   (transit_p = 0.8)
#endif
// per-wave LDS of abd_gibbs_kernel: sort keys u32[G+1] + order u16[G+1] + transit u8[G+1] + log u f64[G+1], each padded to 16 bytes
__host__ __device__ inline size_t abd_gibbs_pad16(size_t b) { return (b + 15) / 16 * 16; }
__host__ __device__ inline size_t abd_gibbs_wave_lds(int G) {
  const size_t n = (size_t)G + 1;
  return abd_gibbs_pad16(4 * n) + abd_gibbs_pad16(2 * n) + abd_gibbs_pad16(n) + abd_gibbs_pad16(8 * n);
}

struct GibbsArgs {
  EvalArgs e;  // panels, packed words, chain parameters (ch[k].rw / waner are updated IN PLACE)
  uint32_t seed_lo, seed_hi, sweep;
  uint32_t ind_offset;  // added to the individual's index in the Philox counter (cohort sharded by individual)
  uint32_t stream[ABD_MAX_BATCH_K];  // third counter word of each chain: its slot id (+ the caller's offset)
  double theta0[ABD_MAX_BATCH_K];  // log p - log(1 - p)             = p_logodds__
  double theta7[ABD_MAX_BATCH_K];  // log p_waner - log(1 - p_waner) = ab_s_p_waner_logodds__
  double is2_n[ABD_MAX_BATCH_K];   // 1 / sigma_n^2
  double is2_s[ABD_MAX_BATCH_K];
  unsigned long long* counts;      // [n_chains][2]: accepted, proposed (integer atomics: order-free)
  unsigned int* work;              // [n_chains]: next individual of each chain (abd_gibbs_dense_kernel's work queue), zeroed per launch
  unsigned long long* stats;       // nullptr, or 8 development counters of abd_gibbs_dense_kernel (ABD_GIBBS_STATS=1)
  int32_t refill_min, tail_lanes, tail_age;  // scheduler knobs of abd_gibbs_dense_kernel (abd_gibbs2.hpp)
};


// scheduler constants of abd_gibbs_dense_kernel (abd_gibbs2.hpp)
#define ABD_G2_REFILL_MIN 16         // idle lanes that trigger a refill (a refill costs ~2-3 walk steps)
#define ABD_G2_TAIL_LANES 8          // walkers left when the whole wave starts finishing them one at a time ...
#define ABD_G2_TAIL_AGE 6            // ... those that have survived this many gaps
