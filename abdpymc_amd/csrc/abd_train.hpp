// abd_train.hpp -- leapfrog trains of dense cohorts: what the LAST workgroup of a train launch (abd_dense.hpp: dense_body,
// DenseTrainArgs) does with the sums of the chains that stepped, and what the launch's service workgroup passes on.
//
// pm.sample runs NUTS on the 17 continuous variables (abd.py:921-922): a chain's leapfrogs follow each other, each needing
// the gradient of the one before.  Here the device takes them one after the other without the host in between: wave k of
// the last workgroup owns chain k of the unit, lane v < 17 of it value variable v, and runs the chain's state machine
// (abd_types.hpp: TrainChain) --
//   * logp and gradient at the point that was evaluated, from the 16 sums and the point's closed-form terms (abd_terms.hpp);
//   * the leapfrog that led there is finished (second half kick), and the next one begun (half kick + drift): the host's
//     arithmetic (abd_nuts.hpp: feed / stage_leapfrog, diagonal metric), operation by operation and without contraction;
//   * when the half of the tree that is being built is complete, the end of the trajectory moves there and the next half
//     starts from the end its pre-drawn direction points away from (abd_nuts.hpp: start_half) -- the host's tree logic (U-turn
//     tests, multinomial sampling, divergences) follows behind on the records and simply stops asking for steps when the
//     tree has ended; what the device took beyond that is never looked at;
//   * the next point, transformed, goes to the other slot of the chain's TrainChain for the launch queued behind this one.
// The step's record (logp, gradient, next point) goes to a ring in mapped host memory: written by this launch when nothing
// may follow (TrainChainArgs::own), else left beside the next point and passed on by the NEXT launch's service workgroup,
// so that no launch of a train waits for its own PCIe writes before its successor may start.
#pragma once

#include "abd_dense.hpp"

// lane v < ABD_NT: the point {t2, ph} of variable v goes to nx, transformed
__device__ __forceinline__ void train_put_point(TrainPoint* nx, int lane, double t2, double ph) {
  const int q4 = lane == 0 ? 0 : lane == 3 ? 1 : lane == 6 ? 2 : lane == 7 ? 3 : -1;
  double tr2, n0, n1;
  abdi::transform_lane(lane, t2, tr2, n0, n1);
  nx->theta[lane] = t2;
  nx->p_half[lane] = ph;
  nx->tr[lane] = tr2;
  if (q4 >= 0) {
    nx->L0[q4] = n0;
    nx->L1[q4] = n1;
  }
}

// half kick + drift from (q, p, g) with signed step ve (abd_nuts.hpp: stage_leapfrog)
__device__ __forceinline__ void train_stage(double q, double p, double g, double ve, double im, double& t2, double& ph) {
  ph = __dadd_rn(p, __dmul_rn(__dmul_rn(0.5, ve), g));  // p_half = cur.p + 0.5 ve cur.g
  t2 = __dadd_rn(q, __dmul_rn(ve, __dmul_rn(im, ph)));  // req_q = cur.q + ve (M^-1 p_half)
}

// the record of step tc.rec_idx / tc.fwd_idx goes to the host: fields first, the tag behind a system-scope fence
__device__ __forceinline__ void train_write_record(TrainRecord* rec, int lane, double lp, double g, double next_theta, double next_p_half,
                                                   double tag) {
  if (lane < ABD_NT) {
    rec->g[lane] = g;
    rec->next_theta[lane] = next_theta;
    rec->next_p_half[lane] = next_p_half;
    if (lane == 0) rec->lp = lp;
  }
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) __hip_atomic_store(&rec->tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the service workgroup: wave k passes chain k's previous record on
__device__ __forceinline__ void train_service(const DenseTrainArgs& a, int wave, int lane) {
  if (wave >= ABD_TRAIN_CB) return;
  const TrainChainArgs& tc = a.tc[wave];
  if (tc.fwd_slot < 0) return;
  const TrainPoint* cur = tc.st->pt + tc.fwd_slot;
  const int k = lane < ABD_NT ? lane : 0;
  train_write_record(tc.ring + (tc.fwd_idx % ABD_TRAIN_RING), lane, cur->prev_lp, cur->prev_g[k], cur->theta[k], cur->p_half[k],
                     (double)(tc.fwd_idx + 1));
}

// one chain of the unit in the launch's last workgroup: sm[0 .. 15] the chain's sums (a step), sm[16 .. 32] scratch
__device__ __forceinline__ void train_step(const DenseTrainArgs& a, const TrainChainArgs& tc, double* sm, int lane) {
  TrainChain* st = tc.st;
  const int k = lane < ABD_NT ? lane : 0;
  if (tc.action == ABD_TR_BEGIN) {
    // a new transition: the host's block (mapped host memory; one PCIe round trip for the wave)
    const TrainBegin* b = tc.begin;
    const double q0 = b->q0[k], p0 = b->p0[k], g0 = b->g0[k], im = b->inv_mass[k];
    const double eps = b->eps;
    const uint32_t dirs = b->dirs;
    const int max_depth = b->max_depth, eval_first = b->eval_first;
    int phase = ABD_PH_IDLE, dir = 1;
    if (lane < ABD_NT) {
      st->inv_mass[lane] = im;
      TrainPoint* nx = st->pt + 0;
      nx->prev_g[lane] = 0.0;
      if (lane == 0) nx->prev_lp = 0.0;
      if (eval_first) {
        // the point itself goes out for evaluation; its momentum waits in p_half (abd_nuts.hpp: begin_draw)
        train_put_point(nx, lane, q0, p0);
        phase = ABD_PH_EVAL0;
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          st->end[e].q[lane] = q0;
          st->end[e].p[lane] = p0;
          st->end[e].g[lane] = g0;
        }
        if (max_depth > 0) {
          dir = dirs & 1u ? 1 : -1;
          double t2, ph;
          train_stage(q0, p0, g0, (double)dir * eps, im, t2, ph);
          train_put_point(nx, lane, t2, ph);
          phase = ABD_PH_LEAF;
        }
      }
    }
    if (lane == 0) {
      st->eps = eps;
      st->dirs = dirs;
      st->max_depth = max_depth;
      st->phase = phase;
      st->depth = 0;
      st->n_leaf = 0;
      st->n_target = 1;
      st->dir = dir;
    }
    return;
  }

  // ---- a step: logp and gradient at the point that was evaluated ----
  const TrainPoint* cur = st->pt + tc.use_slot;
  TrainPoint* nx = st->pt + (tc.use_slot ^ 1);
  double* lps = sm + 16;  // [17] the variables' shares of logp
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
  m.dense = 1;
  m.N = (double)a.N;
  m.cells = (double)a.G * (double)a.N;
  m.Kn = (double)a.K_n;
  m.Ks = (double)a.K_s;
  m.prior_const = a.prior_const;
  const double tk = cur->theta[k];
  const double l0 = q4 >= 0 ? cur->L0[q4] : 0.0, l1 = q4 >= 0 ? cur->L1[q4] : 0.0;
  double lp_k, g_k;
  abdi::assemble_lane(k, m, tr, tk, cur->tr[k], l0, l1, sm, lp_k, g_k);
  if (lane < ABD_NT) lps[lane] = lp_k;
  __builtin_amdgcn_wave_barrier();
  double lp = 0.0;
#pragma unroll
  for (int q = 0; q < ABD_NT; ++q) lp += lps[q];  // every lane, same order

  // ---- the chain's state machine ----
  const int phase = st->phase, depth = st->depth, n_leaf = st->n_leaf, n_target = st->n_target, dir = st->dir, max_depth = st->max_depth;
  const uint32_t dirs = st->dirs;
  const double eps = st->eps, im = st->inv_mass[k];
  const bool finite = __builtin_isfinite(lp);
  const double gd = finite ? g_k : 0.0;  // feed: cur.g = finite ? g1 : 0
  double t2 = 0.0, ph = 0.0;              // the next point
  bool have_next = false;
  int n_phase = ABD_PH_IDLE, n_depth = depth, n_nleaf = n_leaf, n_ntarget = n_target, n_dir = dir;
  if (phase == ABD_PH_EVAL0) {
    // the transition's start point: both ends of the trajectory are here, with the momentum that waited in p_half
    const double p0 = cur->p_half[k];
    if (lane < ABD_NT) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        st->end[e].q[lane] = tk;
        st->end[e].p[lane] = p0;
        st->end[e].g[lane] = gd;
      }
    }
    if (max_depth > 0) {
      n_dir = dirs & 1u ? 1 : -1;
      train_stage(tk, p0, gd, (double)n_dir * eps, im, t2, ph);
      have_next = true;
      n_phase = ABD_PH_LEAF;
      n_depth = 0;
      n_nleaf = 0;
      n_ntarget = 1;
    }
  } else if (phase == ABD_PH_LEAF) {
    const double ve = (double)dir * eps;
    const double kick = __dmul_rn(__dmul_rn(0.5, ve), gd);  // 0.5 * ve * g
    const double p = __dadd_rn(cur->p_half[k], kick);       // feed: cur.p = p_half + 0.5 ve g
    if (n_leaf + 1 == n_target) {
      // the half is complete: this end of the trajectory moves here, the next half starts where its direction points away from
      if (lane < ABD_NT) {
        TrainEnd* e = st->end + (dir > 0 ? 1 : 0);
        e->q[lane] = tk;
        e->p[lane] = p;
        e->g[lane] = gd;
      }
      n_depth = depth + 1;
      if (n_depth < max_depth) {
        n_dir = (dirs >> n_depth) & 1u ? 1 : -1;
        double qe = tk, pe = p, ge = gd;
        if (n_dir != dir) {  // the other end: untouched by this step
          const TrainEnd* o = st->end + (n_dir > 0 ? 1 : 0);
          qe = o->q[k];
          pe = o->p[k];
          ge = o->g[k];
        }
        train_stage(qe, pe, ge, (double)n_dir * eps, im, t2, ph);
        have_next = true;
        n_phase = ABD_PH_LEAF;
        n_nleaf = 0;
        n_ntarget = 1 << n_depth;
      }
    } else {
      ph = __dadd_rn(p, kick);                            // stage_leapfrog: p_half = cur.p + 0.5 ve cur.g
      t2 = __dadd_rn(tk, __dmul_rn(ve, __dmul_rn(im, ph)));  // req_q = cur.q + ve v
      have_next = true;
      n_phase = ABD_PH_LEAF;
      n_nleaf = n_leaf + 1;
    }
  }
  if (lane < ABD_NT) {
    if (have_next) train_put_point(nx, lane, t2, ph);
    nx->prev_g[lane] = g_k;  // the next launch's service workgroup passes this step's record on
    if (lane == 0) nx->prev_lp = lp;
  }
  if (lane == 0) {
    st->phase = n_phase;
    st->depth = n_depth;
    st->n_leaf = n_nleaf;
    st->n_target = n_ntarget;
    st->dir = n_dir;
  }
  if (tc.own) train_write_record(tc.ring + (tc.rec_idx % ABD_TRAIN_RING), lane, lp, g_k, t2, ph, (double)(tc.rec_idx + 1));
}
