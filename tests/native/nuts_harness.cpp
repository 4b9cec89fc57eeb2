// Test harness (CPU only): drives the NUTS state machine of abdpymc_amd/csrc/abd_nuts.hpp against an
// independent-normal target whose moments are known in closed form.  Built by tests/test_nuts_native.py with g++.
#include "abd_nuts.hpp"

// target: independent normals (prec == NULL: mean, sd) or a correlated normal with precision matrix prec (17 x 17)
extern "C" int nuts_harness_run(const double* mean, const double* sd, const double* prec, long long tune, long long draws,
                                unsigned long long seed, int n_chains, int dense, double* out_q, double* out_stats) {
  using namespace abdnuts;
  auto eval = [&](const double* q, double* g) {
    double lp = 0;
    if (prec) {
      for (int r = 0; r < D; ++r) {
        double s = 0;
        for (int c = 0; c < D; ++c) s += prec[r * D + c] * (q[c] - mean[c]);
        g[r] = -s;
        lp -= 0.5 * s * (q[r] - mean[r]);
      }
      return lp;
    }
    for (int d = 0; d < D; ++d) {
      const double z = (q[d] - mean[d]) / sd[d];
      lp -= 0.5 * z * z;
      g[d] = -z / sd[d];
    }
    return lp;
  };
  // chains advance in lock step, one pending evaluation each, the way the device driver batches them
  AdaptiveNuts* ch = new AdaptiveNuts[n_chains];
  for (int c = 0; c < n_chains; ++c) {
    double q0[D], g0[D];
    for (int d = 0; d < D; ++d) q0[d] = mean[d] + sd[d] * (c % 2 ? 1.5 : -1.5);
    const double lp0 = eval(q0, g0);
    ch[c].init(q0, lp0, g0, seed, (unsigned long long)c, tune, 10, 0.8, dense != 0);
  }
  for (long long it = 0; it < tune + draws; ++it) {
    for (int c = 0; c < n_chains; ++c) ch[c].begin();
    bool any = true;
    while (any) {
      any = false;
      for (int c = 0; c < n_chains; ++c) {
        if (!ch[c].nuts.active) continue;
        any = true;
        double g[D];
        const double lp = eval(ch[c].nuts.request(), g);
        ch[c].nuts.feed(lp, g);
      }
    }
    for (int c = 0; c < n_chains; ++c) {
      ch[c].end_transition();
      if (it >= tune) {
        const long long k = it - tune;
        double* q = out_q + ((size_t)c * draws + k) * D;
        for (int d = 0; d < D; ++d) q[d] = ch[c].nuts.q[d];
        double* s = out_stats + ((size_t)c * draws + k) * 6;
        s[0] = ch[c].nuts.stats.lp;
        s[1] = ch[c].nuts.stats.tree_depth;
        s[2] = ch[c].nuts.stats.n_steps;
        s[3] = ch[c].nuts.stats.mean_tree_accept;
        s[4] = ch[c].nuts.stats.step_size;
        s[5] = ch[c].nuts.stats.diverging ? 1.0 : 0.0;
      }
    }
  }
  delete[] ch;
  return 0;
}

// ---- the leapfrog-train protocol (abd_train.hpp <-> abd_sampler.hip: sampler_run_trains), with the DEVICE's part restated on
// the CPU: a per-chain state machine that holds both ends of the tree and the pre-drawn doubling directions, takes one leapfrog
// per step -- across the halves of a doubling and across doublings -- and leaves a record (logp, gradient, next point) for the
// host, whose tree logic follows behind on the records (Nuts::feed(..., across_halves = true)).  Same draws, bit for bit, as
// the classic request / feed loop above: the arithmetic is stage_leapfrog's, operation by operation.
namespace {
struct DeviceChain {  // abd_types.hpp: TrainChain (+ the pending point of its slot)
  double end_q[2][abdnuts::D], end_p[2][abdnuts::D], end_g[2][abdnuts::D], inv_mass[abdnuts::D], eps = 0;
  unsigned dirs = 0;
  int max_depth = 0, phase = 0 /* 0 idle, 1 eval0, 2 leaf */, depth = 0, n_leaf = 0, n_target = 1, dir = 1;
  double pt_q[abdnuts::D], pt_ph[abdnuts::D];
};
struct DeviceRecord {
  double lp, g[abdnuts::D], next_q[abdnuts::D], next_ph[abdnuts::D];
};
void dev_stage(const double* q, const double* p, const double* g, double ve, const double* im, double* t2, double* ph) {
  for (int d = 0; d < abdnuts::D; ++d) {
    ph[d] = p[d] + (0.5 * ve) * g[d];    // train_stage: p_half = cur.p + 0.5 ve cur.g
    t2[d] = q[d] + ve * (im[d] * ph[d]);  //              req_q  = cur.q + ve (M^-1 p_half)
  }
}
// abd_train.hpp: train_step, action BEGIN
void dev_begin(DeviceChain& st, const double* q0, const double* p0, const double* g0, const double* inv_mass, double eps, unsigned dirs,
               int max_depth, bool eval_first) {
  using abdnuts::D;
  std::memcpy(st.inv_mass, inv_mass, sizeof(st.inv_mass));
  st.eps = eps;
  st.dirs = dirs;
  st.max_depth = max_depth;
  st.depth = 0;
  st.n_leaf = 0;
  st.n_target = 1;
  st.dir = 1;
  st.phase = 0;
  if (eval_first) {
    std::memcpy(st.pt_q, q0, sizeof(st.pt_q));
    std::memcpy(st.pt_ph, p0, sizeof(st.pt_ph));
    st.phase = 1;
    return;
  }
  for (int e = 0; e < 2; ++e) {
    std::memcpy(st.end_q[e], q0, sizeof(double) * D);
    std::memcpy(st.end_p[e], p0, sizeof(double) * D);
    std::memcpy(st.end_g[e], g0, sizeof(double) * D);
  }
  if (max_depth > 0) {
    st.dir = dirs & 1u ? 1 : -1;
    dev_stage(q0, p0, g0, (double)st.dir * eps, st.inv_mass, st.pt_q, st.pt_ph);
    st.phase = 2;
  }
}
// abd_train.hpp: train_step, a step (lp, g: the target at the pending point); false when the chain was idle
bool dev_step(DeviceChain& st, double lp, const double* g, DeviceRecord& rec) {
  using abdnuts::D;
  if (st.phase == 0) return false;
  const bool finite = std::isfinite(lp);
  double gd[D], tk[D], t2[D], ph[D];
  for (int d = 0; d < D; ++d) {
    gd[d] = finite ? g[d] : 0.0;
    tk[d] = st.pt_q[d];
    t2[d] = ph[d] = 0.0;
  }
  bool have_next = false;
  if (st.phase == 1) {
    for (int e = 0; e < 2; ++e) {
      std::memcpy(st.end_q[e], tk, sizeof tk);
      std::memcpy(st.end_p[e], st.pt_ph, sizeof tk);
      std::memcpy(st.end_g[e], gd, sizeof tk);
    }
    st.phase = 0;
    if (st.max_depth > 0) {
      st.dir = st.dirs & 1u ? 1 : -1;
      dev_stage(tk, st.pt_ph, gd, (double)st.dir * st.eps, st.inv_mass, t2, ph);
      have_next = true;
      st.phase = 2;
      st.depth = 0;
      st.n_leaf = 0;
      st.n_target = 1;
    }
  } else {
    const double ve = (double)st.dir * st.eps;
    double p[D], kick[D];
    for (int d = 0; d < D; ++d) {
      kick[d] = (0.5 * ve) * gd[d];
      p[d] = st.pt_ph[d] + kick[d];  // feed: cur.p = p_half + 0.5 ve g
    }
    if (st.n_leaf + 1 == st.n_target) {
      const int e = st.dir > 0 ? 1 : 0;
      std::memcpy(st.end_q[e], tk, sizeof tk);
      std::memcpy(st.end_p[e], p, sizeof tk);
      std::memcpy(st.end_g[e], gd, sizeof tk);
      st.depth += 1;
      st.phase = 0;
      if (st.depth < st.max_depth) {
        st.dir = (st.dirs >> st.depth) & 1u ? 1 : -1;
        const int o = st.dir > 0 ? 1 : 0;
        dev_stage(st.end_q[o], st.end_p[o], st.end_g[o], (double)st.dir * st.eps, st.inv_mass, t2, ph);
        have_next = true;
        st.phase = 2;
        st.n_leaf = 0;
        st.n_target = 1 << st.depth;
      }
    } else {
      for (int d = 0; d < D; ++d) {
        ph[d] = p[d] + kick[d];
        t2[d] = tk[d] + ve * (st.inv_mass[d] * ph[d]);
      }
      have_next = true;
      st.n_leaf += 1;
    }
  }
  rec.lp = lp;
  std::memcpy(rec.g, g, sizeof rec.g);
  std::memcpy(rec.next_q, t2, sizeof t2);
  std::memcpy(rec.next_ph, ph, sizeof ph);
  if (have_next) {
    std::memcpy(st.pt_q, t2, sizeof t2);
    std::memcpy(st.pt_ph, ph, sizeof ph);
  }
  return true;
}
}  // namespace

// independent normals, diagonal metric; eval_first: every transition starts with an evaluation of its start point (what the
// compound step does after a sweep: begin_draw / set_point / begin_finish / adopt_request); run_on: steps the device takes
// beyond the end of a tree before the host stops it (the look-ahead: never looked at)
extern "C" int nuts_harness_run_trains(const double* mean, const double* sd, long long tune, long long draws, unsigned long long seed,
                                       int n_chains, int eval_first, int run_on, double* out_q, double* out_stats) {
  using namespace abdnuts;
  auto eval = [&](const double* q, double* g) {
    double lp = 0;
    for (int d = 0; d < D; ++d) {
      const double z = (q[d] - mean[d]) / sd[d];
      lp -= 0.5 * z * z;
      g[d] = -z / sd[d];
    }
    return lp;
  };
  for (int c = 0; c < n_chains; ++c) {
    AdaptiveNuts ch;
    DeviceChain dev;
    double q0[D], g0[D];
    for (int d = 0; d < D; ++d) q0[d] = mean[d] + sd[d] * (c % 2 ? 1.5 : -1.5);
    ch.init(q0, eval(q0, g0), g0, seed, (unsigned long long)c, tune, 10, 0.8, false);
    for (long long it = 0; it < tune + draws; ++it) {
      Nuts& nu = ch.nuts;
      DeviceRecord rec;
      double g[D];
      if (eval_first) {
        ch.begin_draw();
        dev_begin(dev, nu.q, nu.p0_pending, nu.g, nu.inv_mass, nu.eps, nu.dir_bits, nu.max_depth, true);
        const double lp = eval(dev.pt_q, g);
        if (!dev_step(dev, lp, g, rec)) return 2;
        nu.set_point(rec.lp, rec.g);
        nu.begin_finish();
        nu.adopt_request(rec.next_q, rec.next_ph);
      } else {
        ch.begin();
        dev_begin(dev, nu.q, nu.p0_pending, nu.g, nu.inv_mass, nu.eps, nu.dir_bits, nu.max_depth, false);
      }
      int steps = 0;
      while (nu.active) {
        if (std::memcmp(dev.pt_q, nu.request(), sizeof(double) * D) != 0) return 3;  // the device evaluates what the host asks for
        const double lp = eval(dev.pt_q, g);
        if (!dev_step(dev, lp, g, rec)) return 4;  // the device must not run out of steps before the host's tree ends
        nu.feed(rec.lp, rec.g, rec.next_q, rec.next_ph, true);
        if (++steps > nu.max_leaves()) return 5;
      }
      for (int k = 0; k < run_on; ++k) {  // what is queued behind the end of the tree
        const double lp = eval(dev.pt_q, g);
        if (!dev_step(dev, lp, g, rec)) break;
      }
      ch.end_transition();
      if (it >= tune) {
        const long long k = it - tune;
        double* q = out_q + ((size_t)c * draws + k) * D;
        for (int d = 0; d < D; ++d) q[d] = nu.q[d];
        double* s = out_stats + ((size_t)c * draws + k) * 6;
        s[0] = nu.stats.lp;
        s[1] = nu.stats.tree_depth;
        s[2] = nu.stats.n_steps;
        s[3] = nu.stats.mean_tree_accept;
        s[4] = nu.stats.step_size;
        s[5] = nu.stats.diverging ? 1.0 : 0.0;
      }
    }
  }
  return 0;
}
