// CPU harness for abd_terms.hpp (tests/test_terms_native.py): the closed-form part of the joint logp written per value
// variable (assemble_lane: what a leapfrog train's launch runs, one lane per variable) against the serial form every
// fetched evaluation goes through (assemble_terms).
#include "abd_terms.hpp"

extern "C" int terms_compare(const double* theta, const double* sums, int G, double N, double Kn, double Ks, int dense,
                             double* lp_serial, double* g_serial, double* lp_lanes, double* g_lanes) {
  using namespace abdi;
  ModelSizes m;
  m.G = G;
  m.dense = dense;
  m.N = N;
  m.cells = (double)G * N;
  m.Kn = Kn;
  m.Ks = Ks;
  m.prior_const = 0.25;
  const HostTerms h = prepare(theta);
  assemble_terms(m, h, theta, sums, lp_serial, g_serial, true);
  double lp = 0.0;
  const double* trv = &h.tr.p;
  for (int k = 0; k < ABD_N_THETA; ++k) {
    double tr, l0, l1, lpk, gk;
    transform_lane(k, theta[k], tr, l0, l1);
    if (tr != trv[k]) return 1;  // the same backward transform
    const int q4 = k == 0 ? 0 : k == 3 ? 1 : k == 6 ? 2 : k == 7 ? 3 : -1;
    if (q4 >= 0 && (l0 != h.L0[q4] || l1 != h.L1[q4])) return 2;  // the same softplus pair
    assemble_lane(k, m, h.tr, theta[k], trv[k], q4 >= 0 ? h.L0[q4] : 0.0, q4 >= 0 ? h.L1[q4] : 0.0, sums, lpk, gk);
    lp += lpk;
    g_lanes[k] = gk;
  }
  *lp_lanes = lp;
  return 0;
}
