// abd_terms.hpp -- the closed-form part of the joint log-probability: backward transforms of the 17 value variables,
// PyMC's prior densities + transform log-Jacobians, and the combination of the device sums with them.
// One source for the host (abd_context.hip: every evaluation a caller fetches) and for the device (abd_dense.hpp: a
// leapfrog train's launches assemble logp and gradient themselves, so that the next point needs no host round trip).
#pragma once

#include <cmath>

#include "../../include/abd_hip.h"
#include "abd_types.hpp"

#define ABD_HD __host__ __device__ __forceinline__

namespace abdi {

constexpr double kLog2Pi = 1.8378770664093453;  // log(2 pi)

struct Transformed {
  double p, perm_n, temp_n, rho_n, init_n, perm_s, rho_s, q, tinf, tvac, init_s;
  double b_n, d_n, sig_n, b_s, d_s, sig_s;
};
static_assert(sizeof(Transformed) == ABD_N_THETA * sizeof(double), "Transformed is indexed like theta (abd_dense.hpp: train epilogue)");

// see prepare()
struct HostTerms {
  Transformed tr;
  double L0[4], L1[4];  // -softplus(-t), -softplus(t) of theta[0], [3], [6], [7]
};

// what the closed forms need to know about the cohort
struct ModelSizes {
  int32_t G, dense;
  double N, cells, Kn, Ks, prior_const;
};

ABD_HD double sigmoid(double t) { return 1.0 / (1.0 + exp(-t)); }
ABD_HD double softplus(double t) { return fmax(t, 0.0) + log1p(exp(-fabs(t))); }

// the backward transform of value variable k: logodds -> sigmoid, log -> exp, else identity
ABD_HD double transform_one(int k, double t) {
  switch (k) {
    case 0: case 3: case 6: case 7: return sigmoid(t);
    case 1: case 2: case 5: case 8: case 9: case 13: case 16: return exp(t);
    // b = 0 (a flat curve) is replaced by the smallest scale that keeps c = 1024 log2(e) b a normal number: the logistic
    // term is 1/2 to the last bit either way, and the dense kernel's sum for d/db, which it returns scaled by c, stays defined
    case 11: case 14: return t == 0.0 ? 1e-300 : t;
    default: return t;
  }
}

ABD_HD Transformed transform(const double* t) {
  Transformed c;
  double* v = &c.p;
  for (int k = 0; k < ABD_N_THETA; ++k) v[k] = transform_one(k, t[k]);
  return c;
}

// Everything transcendental that one theta needs -- the backward transforms and the softplus pairs of the four
// logit-transformed variables -- computed once, when the evaluation is QUEUED (the host is ahead of the device then), so
// that fetching a result is a few dozen multiply-adds (at config 3 the fetch of a region's results was 4 % of the region).
ABD_HD HostTerms prepare(const double* t) {
  HostTerms h;
  h.tr = transform(t);
  const int k4[4] = {0, 3, 6, 7};
  for (int q = 0; q < 4; ++q) {
    h.L0[q] = -softplus(-t[k4[q]]);
    h.L1[q] = -softplus(t[k4[q]]);
  }
  return h;
}

// Priors + transform log-Jacobians in closed form, and their gradient.
//   p ~ Beta(1, G-1), i_raw ~ Bernoulli(p)                         abd.py:424-427
//   ab_n_perm/temp ~ Gamma, ab_n_rho ~ Beta(10,1), ab_n_init ~ N  abd.py:329-340
//   ab_s_* likewise, ab_s_waner ~ Bernoulli(p_waner)               abd.py:367-388
//   it_*_b ~ N(-1,.5), it_*_d ~ N(2,.5), it_*_sigma ~ Exp(1)       abd.py:464-467
// prior_const: the theta-independent part (abd_context.hip: prior_constant)
ABD_HD double priors(const HostTerms& h, const double* t, int G, double cells, double n1, double N, double m1,
                            double* g /*17 or null*/, double prior_const) {
  double lp = prior_const;
  if (g)
    for (int k = 0; k < ABD_N_THETA; ++k) g[k] = 0.0;
  {  // theta0
    const double L0 = h.L0[0], L1 = h.L1[0], p = h.tr.p;
    const double bm1 = (double)(G - 1) - 1.0;
    lp += (bm1 == 0.0 ? 0.0 : bm1 * L1) + L0 + L1 + n1 * L0 + (cells - n1) * L1;
    if (g) g[0] = (1.0 + n1) * (1.0 - p) - p * (bm1 + 1.0 + (cells - n1));
  }
  const int gk[5] = {1, 2, 5, 8, 9};
  const double gmu[5] = {2.0, 1.0, 2.0, 1.0, 1.0};
  const double gx[5] = {h.tr.perm_n, h.tr.temp_n, h.tr.perm_s, h.tr.tinf, h.tr.tvac};  // exp(t[1]), [2], [5], [8], [9]
  for (int q = 0; q < 5; ++q) {
    const double al = gmu[q] * gmu[q] / (0.5 * 0.5), be = gmu[q] / (0.5 * 0.5);  // Gamma(mu, sd = 0.5): alpha, rate
    const double x = gx[q];
    lp += al * t[gk[q]] - be * x;
    if (g) g[gk[q]] = al - be * x;
  }
  for (int w = 0; w < 2; ++w) {
    const int k = w == 0 ? 3 : 6;
    const double L0 = h.L0[w + 1], L1 = h.L1[w + 1], r = w == 0 ? h.tr.rho_n : h.tr.rho_s;
    lp += 9.0 * L0 + L0 + L1;
    if (g) g[k] = 10.0 * (1.0 - r) - r;
  }
  {
    const double L0 = h.L0[3], L1 = h.L1[3], q = h.tr.q;
    lp += L0 + L1 + m1 * L0 + (N - m1) * L1;
    if (g) g[7] = (1.0 + m1) * (1.0 - q) - q * (1.0 + (N - m1));
  }
  const int nk[6] = {4, 10, 11, 12, 14, 15};
  const double nmu[6] = {-2.0, -2.0, -1.0, 2.0, -1.0, 2.0};
  const double nsd[6] = {1.0, 1.0, 0.5, 0.5, 0.5, 0.5};
  for (int q = 0; q < 6; ++q) {
    const double z = (t[nk[q]] - nmu[q]) / nsd[q];
    lp += -0.5 * z * z;
    if (g) g[nk[q]] = -z / nsd[q];
  }
  for (int w = 0; w < 2; ++w) {
    const int k = w == 0 ? 13 : 16;
    const double x = w == 0 ? h.tr.sig_n : h.tr.sig_s;
    lp += -x + t[k];
    if (g) g[k] = -x + 1.0;
  }
  return lp;
}

// Combine the device sums of one chain with the closed-form terms.  The device accumulates
//   Q2 = sum q^2, H.. = sums of h' = q s (1 - s), QS = sum q s   with q = od - d s   (abd_types.hpp)
// so  ll = -1/2 Q2 / sigma^2 - K (log sigma + 1/2 log 2 pi),  d ll / d a_k = -b (d / sigma^2) h'_k.
ABD_HD void assemble_terms(const ModelSizes& m, const HostTerms& h, const double* t, const double* sums, double* logp,
                                  double* grad, bool with_priors) {
  const Transformed& tr = h.tr;
  const double n1 = sums[ABD_NACC], m1 = sums[ABD_NACC + 1];
  double lp = 0.0;
  if (with_priors)
    lp = priors(h, t, m.G, m.cells, n1, m.N, m1, grad, m.prior_const);
  else if (grad)
    for (int k = 0; k < ABD_N_THETA; ++k) grad[k] = 0.0;
  const double is2_n = 1.0 / (tr.sig_n * tr.sig_n), is2_s = 1.0 / (tr.sig_s * tr.sig_s);
  lp += -0.5 * is2_n * sums[A_N_Q2] - m.Kn * (t[13] + 0.5 * kLog2Pi);
  lp += -0.5 * is2_s * sums[A_S_Q2] - m.Ks * (t[16] + 0.5 * kLog2Pi);
  *logp = lp;
  if (grad) {
    const double fn = -tr.b_n * tr.d_n * is2_n, fs = -tr.b_s * tr.d_s * is2_s;
    grad[1] += fn * tr.perm_n * sums[A_N_HC];
    grad[2] += fn * tr.temp_n * sums[A_N_HU];
    grad[3] += fn * tr.temp_n * tr.rho_n * (1.0 - tr.rho_n) * sums[A_N_HD];
    grad[4] += fn * sums[A_N_H];
    // sum h' (a - x): the dense kernel returns it times c = 1024 log2(e) b (abd_dense.hpp), the list kernels as it is
    const double kC = 1.4426950408889634074 * ABD_EXP2_TAB;
    const double hx_n = m.dense ? sums[A_N_HX] / (kC * tr.b_n) : sums[A_N_HX];
    const double hx_s = m.dense ? sums[A_S_HX] / (kC * tr.b_s) : sums[A_S_HX];
    grad[11] += -tr.d_n * is2_n * hx_n;
    grad[12] += is2_n * sums[A_N_QS];
    grad[13] += is2_n * sums[A_N_Q2] - m.Kn;
    grad[5] += fs * tr.perm_s * sums[A_S_HC];
    grad[6] += fs * tr.rho_s * (1.0 - tr.rho_s) * sums[A_S_HD];
    grad[10] += fs * sums[A_S_H];
    grad[14] += -tr.d_s * is2_s * hx_s;
    grad[15] += is2_s * sums[A_S_QS];
    grad[16] += is2_s * sums[A_S_Q2] - m.Ks;
  }
}

// The theta-derived constants of a ChainPar from the backward transforms `tr` (indexed like theta: Transformed).  ONE list of
// fields for the host (abd_context.hip: chain_par) and for the kernels of a leapfrog train, which take the constants of the
// point their predecessor left in device memory (abd_dense.hpp: dense_body, abd_obs.hpp): a field added to ChainPar is added
// here, or the static_assert below stops the build.
ABD_HD void chain_par_from_tr(ChainPar& p, const double* tr) {
  p.perm_n = tr[1];
  p.temp_n = tr[2];
  p.rho_n = tr[3];
  p.init_n = tr[4];
  p.perm_s = tr[5];
  p.rho_s = tr[6];
  p.init_s = tr[10];
  p.b_n = tr[11];
  p.d_n = tr[12];
  p.b_s = tr[14];
  p.d_s = tr[15];
}
static_assert(sizeof(ChainPar) == 11 * sizeof(double) + 4 * sizeof(void*), "ChainPar changed: review chain_par_from_tr (11 constants) and chain_par (4 pointers)");

// ---- the same closed forms, one value variable per caller: lane k of a wave computes what belongs to theta[k] ----
// (abd_dense.hpp: a leapfrog train's launch assembles its own result; a serial pass over the 17 variables costs one lane
// ~600 dependent fp64 operations, ~2.5 us between two launches of a chain)

// backward transform of theta[k] and, for the four logit-transformed variables, the softplus pair -softplus(-t),
// -softplus(t); one exp(-|t|) and one log1p serve both
ABD_HD void transform_lane(int k, double t, double& tr, double& l0, double& l1) {
  tr = transform_one(k, t);
  const double L = log1p(exp(-fabs(t)));
  l0 = -(fmax(-t, 0.0) + L);
  l1 = -(fmax(t, 0.0) + L);
}

// theta[k]'s share of logp (lp_k: the shares of the 17 variables add up to logp, the theta-independent constant is in
// variable 0's) and d logp / d theta[k].  h, t, sums as in assemble_terms; l0 / l1: the softplus pair of theta[k]
// (variables 0, 3, 6, 7); xk: the variable's own transformed value, (&tr.p)[k] (passed on its own: selecting it from the
// struct by k would make the compiler index a copy of the struct in scratch memory).
ABD_HD void assemble_lane(int k, const ModelSizes& m, const Transformed& tr, double tk, double xk, double l0, double l1,
                          const double* sums, double& lp_k, double& g_k) {
  const double n1 = sums[ABD_NACC], m1 = sums[ABD_NACC + 1];
  const double is2_n = 1.0 / (tr.sig_n * tr.sig_n), is2_s = 1.0 / (tr.sig_s * tr.sig_s);
  const double fn = -tr.b_n * tr.d_n * is2_n, fs = -tr.b_s * tr.d_s * is2_s;
  const double kC = 1.4426950408889634074 * ABD_EXP2_TAB;
  double lp = 0.0, g = 0.0;
  switch (k) {
    case 0: {  // p ~ Beta(1, G - 1), i_raw ~ Bernoulli(p)
      const double bm1 = (double)(m.G - 1) - 1.0;
      lp = m.prior_const + (bm1 == 0.0 ? 0.0 : bm1 * l1) + l0 + l1 + n1 * l0 + (m.cells - n1) * l1;
      g = (1.0 + n1) * (1.0 - xk) - xk * (bm1 + 1.0 + (m.cells - n1));
      break;
    }
    case 1: case 2: case 5: case 8: case 9: {  // Gamma(mu, sd = 0.5) on exp(t)
      const double mu = (k == 1 || k == 5) ? 2.0 : 1.0;
      const double al = mu * mu / (0.5 * 0.5), be = mu / (0.5 * 0.5);
      lp = al * tk - be * xk;
      g = al - be * xk;
      if (k == 1) g += fn * xk * sums[A_N_HC];
      if (k == 2) g += fn * xk * sums[A_N_HU];
      if (k == 5) g += fs * xk * sums[A_S_HC];
      break;
    }
    case 3: case 6: {  // Beta(10, 1) on sigmoid(t)
      const double r = xk;
      lp = 9.0 * l0 + l0 + l1;
      g = 10.0 * (1.0 - r) - r;
      g += k == 3 ? fn * tr.temp_n * r * (1.0 - r) * sums[A_N_HD] : fs * r * (1.0 - r) * sums[A_S_HD];
      break;
    }
    case 7:  // p_waner ~ Beta(1, 1), ab_s_waner ~ Bernoulli(p_waner)
      lp = l0 + l1 + m1 * l0 + (m.N - m1) * l1;
      g = (1.0 + m1) * (1.0 - xk) - xk * (1.0 + (m.N - m1));
      break;
    case 4: case 10: case 11: case 12: case 14: case 15: {  // Normal(mu, sd)
      const double mu = (k == 4 || k == 10) ? -2.0 : (k == 11 || k == 14) ? -1.0 : 2.0;
      const double sd = (k == 4 || k == 10) ? 1.0 : 0.5;
      const double z = (tk - mu) / sd;
      lp = -0.5 * z * z;
      g = -z / sd;
      if (k == 4) g += fn * sums[A_N_H];
      if (k == 10) g += fs * sums[A_S_H];
      if (k == 11) g += -tr.d_n * is2_n * (m.dense ? sums[A_N_HX] / (kC * tr.b_n) : sums[A_N_HX]);
      if (k == 14) g += -tr.d_s * is2_s * (m.dense ? sums[A_S_HX] / (kC * tr.b_s) : sums[A_S_HX]);
      if (k == 12) g += is2_n * sums[A_N_QS];
      if (k == 15) g += is2_s * sums[A_S_QS];
      break;
    }
    default: {  // 13, 16: sigma ~ Exponential(1) on exp(t), and the observed Normals' own terms
      const bool nn = k == 13;
      const double x = xk, is2 = nn ? is2_n : is2_s, q2 = nn ? sums[A_N_Q2] : sums[A_S_Q2], K = nn ? m.Kn : m.Ks;
      lp = (-x + tk) + (-0.5 * is2 * q2 - K * (tk + 0.5 * kLog2Pi));
      g = (-x + 1.0) + (is2 * q2 - K);
      break;
    }
  }
  lp_k = lp;
  g_k = g;
}

}  // namespace abdi
