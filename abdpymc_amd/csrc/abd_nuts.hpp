// abd_nuts.hpp -- host side of the compound step: a No-U-Turn transition written as a state machine.
//
// The reference hands the model to pm.sample (abd.py:921-922), which assigns NUTS to the 17 continuous
// value variables.  PyMC's NUTS is a recursive tree builder that calls logp_dlogp once per leapfrog; here
// the same transition is unrolled into "give me the next point to evaluate" / "here is logp and gradient",
// so that a driver can collect the pending points of many chains and evaluate them in ONE device launch.
//
// Algorithm: multinomial NUTS with a diagonal metric (Betancourt 2017; the variant PyMC and Stan run):
//  * tree doubling in a random direction, the new half built leaf by leaf;
//  * every leaf z gets weight exp(H0 - H(z)); the new half's candidate is drawn leaf by leaf with
//    probability w / (sum of w so far) (uniform-multinomial inside a half), and replaces the tree's candidate
//    with probability min(1, W_new / W_old) (biased progressive sampling across halves);
//  * the generalised U-turn test  rho . M^-1 p_- > 0  and  rho . M^-1 p_+ > 0  on every balanced sub-tree of
//    the new half (checked without recursion from O(depth) checkpoints: a leaf with an even index stores
//    (p, running rho); a leaf with an odd index closes one sub-tree per trailing 1-bit of its index) and on
//    the whole tree after each doubling;
//  * a leaf whose energy error exceeds 1000 marks the transition divergent and ends it.
// Step size: dual averaging (Hoffman & Gelman 2014, eq. 6; PyMC's constants gamma 0.05, t0 10, kappa 0.75,
// initial step 0.25 / d^(1/4), mu = log(10 * initial)).  Metric: running weighted variance of the tuning
// draws with a foreground and a background window of 101 draws (PyMC's QuadPotentialDiagAdapt; initial
// variance 1 with weight 10).  Tree depth is capped at 8 during the first 200 tuning draws (PyMC's early_max_treedepth).  PyMC itself is not importable offline: these constants are restated from its
// documentation and the tests check the sampler on closed-form targets, not against PyMC draws.
//
// Optional dense metric (PyMC: init="adapt_full"): M^-1 = regularised covariance of the tuning draws, refreshed at
// the end of windows that double in length; momenta p = L^-T z with M^-1 = L L^T.  This model's posterior is
// strongly correlated (init and perm of an antigen trade off almost exactly), so the dense metric cuts the tree
// depth from ~6 to ~3; the default stays diagonal, as pm.sample's is.
//
// No HIP in this file: tests/native compiles it with g++ against an analytic target.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace abdnuts {

constexpr int D = 17;        // continuous value variables
constexpr int MAX_DEPTH = 16;  // hard cap on the tree depth the checkpoints are sized for

// xoshiro256++ (Blackman & Vigna), seeded through splitmix64: one independent stream per chain
struct Rng {
  uint64_t s[4];
  bool have_spare = false;
  double spare = 0.0;
  static uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  void seed(uint64_t seed, uint64_t stream) {
    uint64_t x = seed ^ (0xD1B54A32D192ED03ull * (stream + 1));
    for (int k = 0; k < 4; ++k) s[k] = splitmix(x);
    have_spare = false;
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return (double)(next() >> 11) * 0x1.0p-53; }  // [0, 1)
  double normal() {                                                  // Box-Muller, both values used
    if (have_spare) {
      have_spare = false;
      return spare;
    }
    const double u = 1.0 - uniform();  // (0, 1]
    const double v = uniform();
    const double r = std::sqrt(-2.0 * std::log(u));
    const double a = 6.283185307179586476925 * v;
    spare = r * std::sin(a);
    have_spare = true;
    return r * std::cos(a);
  }
};

inline double log_add_exp(double a, double b) {
  if (a == -std::numeric_limits<double>::infinity()) return b;
  if (b == -std::numeric_limits<double>::infinity()) return a;
  const double m = a > b ? a : b;
  return m + std::log(std::exp(a - m) + std::exp(b - m));
}

struct DualAveraging {
  double mu = 0, target = 0.8, gamma = 0.05, t0 = 10.0, kappa = 0.75;
  double h_bar = 0, log_eps_bar = 0;
  int64_t m = 0;
  void init(double eps0, double target_accept) {
    mu = std::log(10.0 * eps0);
    target = target_accept;
    h_bar = 0;
    log_eps_bar = 0;
    m = 0;
  }
  double update(double accept_stat) {  // -> step size for the next tuning transition
    m += 1;
    const double w = 1.0 / ((double)m + t0);
    h_bar = (1.0 - w) * h_bar + w * (target - accept_stat);
    const double log_eps = mu - std::sqrt((double)m) / gamma * h_bar;
    const double eta = std::pow((double)m, -kappa);
    log_eps_bar = eta * log_eps + (1.0 - eta) * log_eps_bar;
    return std::exp(log_eps);
  }
  double final_eps() const { return std::exp(log_eps_bar); }
};

// running mean / sum of squared deviations (Welford) with an optional prior weight
struct WeightedVariance {
  double n = 0;
  double mean[D];
  double m2[D];
  void reset() {
    n = 0;
    for (int k = 0; k < D; ++k) mean[k] = m2[k] = 0.0;
  }
  void prior(const double* mean0, double var0, double weight) {
    n = weight;
    for (int k = 0; k < D; ++k) {
      mean[k] = mean0[k];
      m2[k] = var0 * weight;
    }
  }
  void add(const double* x) {
    n += 1.0;
    for (int k = 0; k < D; ++k) {
      const double d = x[k] - mean[k];
      mean[k] += d / n;
      m2[k] += d * (x[k] - mean[k]);
    }
  }
  void variance(double* out) const {
    for (int k = 0; k < D; ++k) out[k] = m2[k] / n;
  }
};

struct MassAdapt {
  WeightedVariance fg, bg;
  int64_t n_seen = 0;
  int window = 101;
  void init(const double* q0) {
    fg.prior(q0, 1.0, 10.0);
    bg.reset();
    n_seen = 0;
  }
  // one tuning draw; writes the new diagonal of M^-1
  void update(const double* q, double* inv_mass) {
    fg.add(q);
    bg.add(q);
    fg.variance(inv_mass);
    for (int k = 0; k < D; ++k)
      if (!(inv_mass[k] > 1e-12) || !std::isfinite(inv_mass[k])) inv_mass[k] = 1e-12;
    n_seen += 1;
    if (n_seen % window == 0) {
      fg = bg;
      bg.reset();
    }
  }
};

// running mean / scatter matrix for the dense metric
struct Covariance {
  double n = 0;
  double mean[D];
  double m2[D][D];
  void reset() {
    n = 0;
    for (int r = 0; r < D; ++r) {
      mean[r] = 0;
      for (int c = 0; c < D; ++c) m2[r][c] = 0;
    }
  }
  void add(const double* x) {
    n += 1.0;
    double d0[D];
    for (int k = 0; k < D; ++k) {
      d0[k] = x[k] - mean[k];
      mean[k] += d0[k] / n;
    }
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) m2[r][c] += d0[r] * (x[c] - mean[c]);
  }
  // Stan's shrinkage: (n / (n + 5)) cov + 1e-3 (5 / (n + 5)) I
  void regularised(double out[D][D]) const {
    const double w = n / (n + 5.0), e = 1e-3 * 5.0 / (n + 5.0);
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) out[r][c] = w * (n > 1 ? m2[r][c] / (n - 1.0) : 0.0) + (r == c ? e : 0.0);
  }
};

struct Stats {
  double lp = 0, energy = 0, step_size = 0, mean_tree_accept = 0, max_energy_error = 0;
  int tree_depth = 0, n_steps = 0;
  bool diverging = false, reached_max_depth = false;
};

struct Phase {
  double q[D], p[D], g[D];
};

struct Nuts {
  // ---- persistent state of the chain ----
  double q[D], g[D], lp = 0;
  double inv_mass[D];   // diagonal of M^-1
  bool dense = false;   // use cov / chol instead of inv_mass
  double cov[D][D];     // M^-1 (dense)
  double chol[D][D];    // lower L with M^-1 = L L^T
  double eps = 0.1;
  int max_depth = 10;
  Rng rng;
  Stats stats;

  // ---- one transition ----
  bool active = false;
  double h0 = 0;
  Phase left, right, cur;
  double rho[D], log_w = 0;
  double prop_q[D], prop_g[D], prop_lp = 0;
  int depth = 0, dir = 1, n_leaf = 0, n_target = 1;
  // the directions of ALL doublings of a transition are drawn when it begins (bit d: the doubling at depth d goes forward):
  // a device that takes a chain's leapfrogs one after the other (abd_train.hpp) can then go on into the next half by
  // itself, while the host's tree logic follows behind on the records
  uint32_t dir_bits = 0;
  double p0_pending[D];  // begin_draw .. begin_finish: the momentum drawn for a transition whose start point is still out for evaluation
  double sub_rho[D], sub_log_w = 0;
  double sub_q[D], sub_g[D], sub_lp = 0;
  double p_ckpt[MAX_DEPTH][D], rho_ckpt[MAX_DEPTH][D];
  double sum_alpha = 0, max_err = 0;
  int n_alpha = 0;
  double req_q[D], p_half[D];

  void init(const double* q0, double lp0, const double* g0, uint64_t seed, uint64_t stream) {
    std::memcpy(q, q0, sizeof(q));
    std::memcpy(g, g0, sizeof(g));
    lp = lp0;
    for (int k = 0; k < D; ++k) inv_mass[k] = 1.0;
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) cov[r][c] = chol[r][c] = r == c ? 1.0 : 0.0;
    rng.seed(seed, stream);
    active = false;
  }
  // install a dense M^-1; returns false (and keeps the old one) if it is not positive definite
  bool set_dense_metric(const double m[D][D]) {
    double l[D][D];
    for (int r = 0; r < D; ++r)
      for (int c = 0; c <= r; ++c) {
        double v = m[r][c];
        for (int k = 0; k < c; ++k) v -= l[r][k] * l[c][k];
        if (r == c) {
          if (!(v > 0.0) || !std::isfinite(v)) return false;
          l[r][c] = std::sqrt(v);
        } else {
          l[r][c] = v / l[c][c];
        }
      }
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) {
        cov[r][c] = m[r][c];
        chol[r][c] = c <= r ? l[r][c] : 0.0;
      }
    for (int k = 0; k < D; ++k) inv_mass[k] = m[k][k];
    dense = true;
    return true;
  }
  void velocity(const double* p, double* v) const {  // M^-1 p
    if (!dense) {
      for (int d = 0; d < D; ++d) v[d] = inv_mass[d] * p[d];
      return;
    }
    for (int r = 0; r < D; ++r) {
      double s = 0;
      for (int c = 0; c < D; ++c) s += cov[r][c] * p[c];
      v[r] = s;
    }
  }
  void set_point(double lp0, const double* g0) {  // the discrete state changed under the chain
    lp = lp0;
    std::memcpy(g, g0, sizeof(g));
  }

  double kinetic(const double* p) const {
    double v[D], k = 0;
    velocity(p, v);
    for (int d = 0; d < D; ++d) k += v[d] * p[d];
    return 0.5 * k;
  }
  double dot_sharp(const double* r, const double* p) const {  // r . M^-1 p
    double v[D], s = 0;
    velocity(p, v);
    for (int d = 0; d < D; ++d) s += r[d] * v[d];
    return s;
  }
  void draw_momentum(double* p) {  // p ~ N(0, M)
    if (!dense) {
      for (int d = 0; d < D; ++d) p[d] = rng.normal() / std::sqrt(inv_mass[d]);
      return;
    }
    double z[D];
    for (int d = 0; d < D; ++d) z[d] = rng.normal();
    for (int r = D - 1; r >= 0; --r) {  // L^T p = z
      double v = z[r];
      for (int c = r + 1; c < D; ++c) v -= chol[c][r] * p[c];
      p[r] = v / chol[r][r];
    }
  }

  // begin() in two parts, for a driver that has to evaluate the start point first (the discrete state changed under the
  // chain: abd.py:922's compound step): begin_draw() draws what the transition needs from the random stream,
  // begin_finish() starts the tree once lp and g at q are known (set_point)
  void begin_draw() {
    if (max_depth > MAX_DEPTH) max_depth = MAX_DEPTH;
    draw_momentum(p0_pending);
    dir_bits = (uint32_t)(rng.next() >> 32);
  }
  void begin() {
    begin_draw();
    begin_finish();
  }
  void begin_finish() {
    double p0[D];
    std::memcpy(p0, p0_pending, sizeof(p0));
    h0 = -lp + kinetic(p0);
    std::memcpy(left.q, q, sizeof(q));
    std::memcpy(left.p, p0, sizeof(p0));
    std::memcpy(left.g, g, sizeof(g));
    right = left;
    std::memcpy(rho, p0, sizeof(p0));
    log_w = 0.0;
    std::memcpy(prop_q, q, sizeof(q));
    std::memcpy(prop_g, g, sizeof(g));
    prop_lp = lp;
    depth = 0;
    sum_alpha = 0;
    n_alpha = 0;
    max_err = 0;
    stats = Stats();
    stats.step_size = eps;
    active = true;
    start_half();
  }

  // the point whose logp and gradient the transition needs next (valid while active)
  const double* request() const { return req_q; }

  // leapfrogs the current half still needs after the one that is out for evaluation: the points of a half follow each
  // other deterministically (stage_leapfrog), so a driver may have the DEVICE take them one after the other (a leapfrog
  // train, abd_dense.hpp) and feed the results as they arrive; a half that ends early simply never asks for the rest
  int half_remaining() const { return active ? n_target - n_leaf - 1 : 0; }
  double signed_step() const { return dir * eps; }
  const double* staged_momentum() const { return p_half; }

  // next_q / next_p_half: the next point of the half and its half-kicked momentum as the device computed them (the same
  // operations as stage_leapfrog, diagonal metric); adopted instead of the host's own, so that what the chain records is
  // exactly what was evaluated
  // across_halves: the device also went on into the next half by itself (abd_train.hpp: it knows dir_bits), so its next
  // point is adopted there too
  void feed(double lp1, const double* g1, const double* next_q = nullptr, const double* next_p_half = nullptr, bool across_halves = false) {
    const int half_before = depth;
    feed_(lp1, g1);
    if (next_q && active && (across_halves || (depth == half_before && n_leaf > 0))) {
      std::memcpy(req_q, next_q, sizeof(req_q));
      std::memcpy(p_half, next_p_half, sizeof(p_half));
    }
  }
  // the first point of the transition as the device staged it (see feed)
  void adopt_request(const double* next_q, const double* next_p_half) {
    std::memcpy(req_q, next_q, sizeof(req_q));
    std::memcpy(p_half, next_p_half, sizeof(p_half));
  }
  // leapfrogs a transition that starts now can take at most
  int max_leaves() const { return (1 << max_depth) - 1; }

 private:
  void feed_(double lp1, const double* g1) {
    const bool finite = std::isfinite(lp1);
    const double ve = dir * eps;
    std::memcpy(cur.q, req_q, sizeof(req_q));
    for (int d = 0; d < D; ++d) {
      cur.g[d] = finite ? g1[d] : 0.0;
      cur.p[d] = p_half[d] + 0.5 * ve * cur.g[d];
    }
    double dh = finite ? (-lp1 + kinetic(cur.p)) - h0 : std::numeric_limits<double>::infinity();
    if (std::isnan(dh)) dh = std::numeric_limits<double>::infinity();
    sum_alpha += dh <= 0 ? 1.0 : std::exp(-dh);
    n_alpha += 1;
    if (std::fabs(dh) > std::fabs(max_err)) max_err = dh;
    if (dh > 1000.0) {
      stats.diverging = true;
      finish();
      return;
    }
    const double w = -dh;
    sub_log_w = log_add_exp(sub_log_w, w);
    if (rng.uniform() < std::exp(w - sub_log_w)) {
      std::memcpy(sub_q, cur.q, sizeof(sub_q));
      std::memcpy(sub_g, cur.g, sizeof(sub_g));
      sub_lp = lp1;
    }
    for (int d = 0; d < D; ++d) sub_rho[d] += cur.p[d];
    bool turning = false;
    const unsigned n = (unsigned)n_leaf;
    const int idx_max = __builtin_popcount(n >> 1);
    if ((n & 1u) == 0) {
      std::memcpy(p_ckpt[idx_max], cur.p, sizeof(cur.p));
      std::memcpy(rho_ckpt[idx_max], sub_rho, sizeof(sub_rho));
    } else {
      const int closing = __builtin_ctz(~n);  // trailing 1-bits: sub-trees that end at this leaf
      for (int k = idx_max; k > idx_max - closing && !turning; --k) {
        double span[D];
        for (int d = 0; d < D; ++d) span[d] = sub_rho[d] - rho_ckpt[k][d] + p_ckpt[k][d];
        turning = !(dot_sharp(span, p_ckpt[k]) > 0.0 && dot_sharp(span, cur.p) > 0.0);
      }
    }
    n_leaf += 1;
    if (turning) {  // the new half is discarded
      depth += 1;
      finish();
      return;
    }
    if (n_leaf < n_target) {
      stage_leapfrog();
      return;
    }
    // the half is complete: merge it into the tree
    if (sub_log_w > log_w || rng.uniform() < std::exp(sub_log_w - log_w)) {
      std::memcpy(prop_q, sub_q, sizeof(sub_q));
      std::memcpy(prop_g, sub_g, sizeof(sub_g));
      prop_lp = sub_lp;
    }
    log_w = log_add_exp(log_w, sub_log_w);
    for (int d = 0; d < D; ++d) rho[d] += sub_rho[d];
    (dir < 0 ? left : right) = cur;
    depth += 1;
    const bool whole_turning = !(dot_sharp(rho, left.p) > 0.0 && dot_sharp(rho, right.p) > 0.0);
    if (whole_turning) {
      finish();
    } else if (depth >= max_depth) {
      stats.reached_max_depth = true;
      finish();
    } else {
      start_half();
    }
  }

  void start_half() {
    dir = (dir_bits >> depth) & 1u ? 1 : -1;
    cur = dir < 0 ? left : right;
    n_leaf = 0;
    n_target = 1 << depth;
    for (int d = 0; d < D; ++d) sub_rho[d] = 0.0;
    sub_log_w = -std::numeric_limits<double>::infinity();
    stage_leapfrog();
  }
  void stage_leapfrog() {  // half kick + drift; the second half kick waits for the gradient
    const double ve = dir * eps;
    double v[D];
    for (int d = 0; d < D; ++d) p_half[d] = cur.p[d] + 0.5 * ve * cur.g[d];
    velocity(p_half, v);
    for (int d = 0; d < D; ++d) req_q[d] = cur.q[d] + ve * v[d];
  }
  void finish() {
    std::memcpy(q, prop_q, sizeof(q));
    std::memcpy(g, prop_g, sizeof(g));
    lp = prop_lp;
    stats.lp = lp;
    stats.tree_depth = depth;
    stats.n_steps = n_alpha;
    stats.mean_tree_accept = n_alpha ? sum_alpha / n_alpha : 0.0;
    stats.max_energy_error = max_err;
    stats.energy = h0;
    active = false;
  }
};

// NUTS + its two adaptations for one chain
struct AdaptiveNuts {
  Nuts nuts;
  DualAveraging da;
  MassAdapt mass;
  bool dense_metric = false;
  Covariance cov_win;  // draws of the current slow window
  int64_t slow_begin = 0, slow_end = 0, win_end = 0, win_len = 0;
  double target = 0.8;
  int64_t tune = 0, it = 0;
  int max_depth_full = 10, early_max_depth = 8;  // PyMC: early_max_treedepth = 8 during the first 200 tuning draws
  void init(const double* q0, double lp0, const double* g0, uint64_t seed, uint64_t stream, int64_t n_tune,
            int max_depth, double target_accept, bool dense = false) {
    nuts.init(q0, lp0, g0, seed, stream);
    nuts.max_depth = max_depth;
    max_depth_full = max_depth;
    nuts.eps = 0.25 / std::pow((double)D, 0.25);
    target = target_accept;
    da.init(nuts.eps, target_accept);
    mass.init(q0);
    dense_metric = dense;
    cov_win.reset();
    // Stan's warm-up schedule: a fast interval (step size only, 7.5 %), slow windows that double (the first
    // 2.5 %; each window's draws alone give the next metric: the early ones are still travelling to the
    // typical set and must not linger in the estimate), a final fast interval (5 %)
    slow_begin = n_tune * 75 / 1000;
    slow_end = n_tune - n_tune * 50 / 1000;
    win_len = n_tune * 25 / 1000 > 10 ? n_tune * 25 / 1000 : 10;
    win_end = slow_begin + win_len;
    if (win_end + 2 * win_len > slow_end) win_end = slow_end;  // no room for a second window: stretch this one
    tune = n_tune;
    it = 0;
  }
  bool tuning() const { return it < tune; }
  // start the transition of iteration `it`
  void begin() {
    begin_draw();
    nuts.begin_finish();
  }
  // ... in two parts (Nuts::begin_draw / begin_finish), for a driver that evaluates the start point in between
  void begin_draw() {
    nuts.max_depth = (it < tune && it < 200 && early_max_depth < max_depth_full) ? early_max_depth : max_depth_full;
    nuts.begin_draw();
  }
  // call when the transition of iteration `it` has finished
  void end_transition() {
    if (it < tune) {
      nuts.eps = da.update(nuts.stats.mean_tree_accept);
      if (!dense_metric) {
        // PyMC stops adapting the metric for the last stretch of tuning so the step size can settle
        if (it < tune - tune / 10 - 1) mass.update(nuts.q, nuts.inv_mass);
      } else {
        if (!nuts.dense && it < slow_end) mass.update(nuts.q, nuts.inv_mass);  // diagonal until the first window closes
        if (it >= slow_begin && it < slow_end) {
          cov_win.add(nuts.q);
          if (it + 1 == win_end) {
            double m[D][D];
            cov_win.regularised(m);
            if (cov_win.n > D && nuts.set_dense_metric(m)) da.init(nuts.eps, target);  // step size restarts
            cov_win.reset();
            win_len *= 2;
            win_end += win_len;
            if (win_end + 2 * win_len > slow_end) win_end = slow_end;
          }
        }
      }
      if (it == tune - 1) nuts.eps = da.final_eps();
    }
    it += 1;
  }
};

}  // namespace abdnuts
