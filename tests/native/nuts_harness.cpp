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
