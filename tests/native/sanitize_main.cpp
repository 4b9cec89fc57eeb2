// AddressSanitizer / UBSan driver for the HIP-free host code (CPU only; the GPU side is never sanitised):
//   * the NUTS state machine of the native sampler (abdpymc_amd/csrc/abd_nuts.hpp) through tests/native/nuts_harness.cpp,
//     classic loop and leapfrog-train protocol
//   * the plain-C restatement oracle/abd_oracle.c: logp + gradient (dense and ragged observation lists, 0-2 splits,
//     ignore_pcrpos) and the Gibbs sweep
// Built and run by tests/test_sanitizers.py with -fsanitize=address,undefined -fno-sanitize-recover=all.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" int nuts_harness_run(const double* mean, const double* sd, const double* prec, long long tune, long long draws,
                                unsigned long long seed, int n_chains, int dense, double* out_q, double* out_stats);
extern "C" int nuts_harness_run_trains(const double* mean, const double* sd, long long tune, long long draws, unsigned long long seed,
                                       int n_chains, int eval_first, int run_on, double* out_q, double* out_stats);
extern "C" int abd_oracle_logp_dlogp(int G, int N, int n_splits, const int* splits, const int8_t* vacs, const int8_t* pcrpos,
                                     int64_t K_s, const int32_t* s_gap, const int32_t* s_ind, const double* s_x,
                                     const double* s_y, int64_t K_n, const int32_t* n_gap, const int32_t* n_ind,
                                     const double* n_x, const double* n_y, const int8_t* i_raw, const int8_t* waner,
                                     const double* theta, double* logp, double* grad, int8_t* i_out, int nthreads);
extern "C" int abd_oracle_gibbs_sweep(int G, int N, int n_splits, const int* splits, const int8_t* vacs, const int8_t* pcrpos,
                                      int64_t K_s, const int32_t* s_gap, const int32_t* s_ind, const double* s_x,
                                      const double* s_y, int64_t K_n, const int32_t* n_gap, const int32_t* n_ind,
                                      const double* n_x, const double* n_y, int8_t* i_raw, int8_t* waner, const double* theta,
                                      int chain, uint64_t seed, uint32_t sweep, int64_t* accepted, int64_t* proposed, int nthreads);

static uint64_t rng_state = 88172645463325252ull;
static double urand() {  // xorshift64*
  rng_state ^= rng_state >> 12;
  rng_state ^= rng_state << 25;
  rng_state ^= rng_state >> 27;
  return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

#define REQUIRE(cond)                                                  \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #cond); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

static int run_oracle(int G, int N, int n_splits, const int* splits, bool dense, bool ignore_pcr) {
  std::vector<int8_t> vacs((size_t)N * G), pcr((size_t)N * G), i_raw((size_t)G * N), waner((size_t)N), i_out((size_t)G * N);
  for (auto& v : vacs) v = urand() < 1.5 / G;
  for (auto& v : pcr) v = urand() < 1.0 / G;
  for (auto& v : i_raw) v = urand() < 2.0 / G;
  for (auto& v : waner) v = urand() < 0.5;
  std::vector<int32_t> gap[2], ind[2];
  std::vector<double> x[2], y[2];
  for (int a = 0; a < 2; ++a) {
    if (dense) {
      for (int g = 0; g < G; ++g)
        for (int j = 0; j < N; ++j) {
          gap[a].push_back(g);
          ind[a].push_back(j);
        }
    } else {
      const int K = a == 0 ? 3 * N : 0;  // ragged lists with repeats; the second antigen is EMPTY
      for (int k = 0; k < K; ++k) {
        gap[a].push_back((int32_t)(urand() * G) % G);
        ind[a].push_back((int32_t)(urand() * N) % N);
      }
    }
    for (size_t k = 0; k < gap[a].size(); ++k) {
      x[a].push_back(2.0 * (int)(urand() * 3));
      y[a].push_back(2.0 * urand());
    }
  }
  double theta[17] = {-4.0, 0.7, 0.0, 2.3, -2.0, 0.7, 2.3, 0.0, 0.0, 0.0, -2.0, -1.0, 2.0, -1.0, -1.0, 2.0, -1.0};
  for (double& t : theta) t += 0.2 * (urand() - 0.5);
  double lp = 0.0, grad[17];
  const int8_t* pp = ignore_pcr ? nullptr : pcr.data();
  for (int threads : {1, 3}) {
    int rc = abd_oracle_logp_dlogp(G, N, n_splits, splits, vacs.data(), pp, (int64_t)gap[0].size(), gap[0].data(), ind[0].data(),
                                   x[0].data(), y[0].data(), (int64_t)gap[1].size(), gap[1].data(), ind[1].data(), x[1].data(),
                                   y[1].data(), i_raw.data(), waner.data(), theta, &lp, grad, i_out.data(), threads);
    REQUIRE(rc == 0 && std::isfinite(lp));
    for (double g : grad) REQUIRE(std::isfinite(g));
  }
  int64_t acc = -1, prop = -1;
  int rc = abd_oracle_gibbs_sweep(G, N, n_splits, splits, vacs.data(), pp, (int64_t)gap[0].size(), gap[0].data(), ind[0].data(),
                                  x[0].data(), y[0].data(), (int64_t)gap[1].size(), gap[1].data(), ind[1].data(), x[1].data(),
                                  y[1].data(), i_raw.data(), waner.data(), theta, 1, 12345u, 2u, &acc, &prop, 2);
  REQUIRE(rc == 0 && prop > 0 && acc >= 0 && acc <= prop);
  for (int8_t v : i_raw) REQUIRE(v == 0 || v == 1);
  // bad sizes are refused, not read
  REQUIRE(abd_oracle_logp_dlogp(1, N, 0, nullptr, vacs.data(), pp, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                                nullptr, i_raw.data(), waner.data(), theta, &lp, grad, nullptr, 1) == -1);
  return 0;
}

int main() {
  // ---- NUTS: diagonal and dense metric, short runs (every adaptation window boundary is crossed) ----
  const int D = 17;
  std::vector<double> mean(D), sd(D), prec((size_t)D * D, 0.0);
  for (int k = 0; k < D; ++k) {
    mean[k] = 3.0 * (urand() - 0.5);
    sd[k] = std::exp(4.0 * (urand() - 0.5));
    prec[(size_t)k * D + k] = 1.0 / (sd[k] * sd[k]);
  }
  for (int k = 0; k + 1 < D; ++k) {  // a tridiagonal, diagonally dominant precision
    const double c = 0.3 * std::sqrt(prec[(size_t)k * D + k] * prec[(size_t)(k + 1) * D + k + 1]);
    prec[(size_t)k * D + k + 1] = prec[(size_t)(k + 1) * D + k] = c;
  }
  for (int dense = 0; dense < 2; ++dense) {
    const long long tune = 300, draws = 200;
    const int chains = 2;
    std::vector<double> q((size_t)chains * draws * D), st((size_t)chains * draws * 6);
    int rc = nuts_harness_run(mean.data(), sd.data(), dense ? prec.data() : nullptr, tune, draws, 7ull, chains, dense, q.data(), st.data());
    REQUIRE(rc == 0);
    for (double v : q) REQUIRE(std::isfinite(v));
    if (!dense) {  // the leapfrog-train protocol (device state machine restated on the CPU): the same draws
      std::vector<double> q2(q.size()), st2(st.size());
      for (int eval_first = 0; eval_first < 2; ++eval_first) {
        rc = nuts_harness_run_trains(mean.data(), sd.data(), tune, draws, 7ull, chains, eval_first, 4, q2.data(), st2.data());
        REQUIRE(rc == 0);
        REQUIRE(q2 == q);
      }
    }
  }
  // ---- oracle: shapes the reference's tests and BASELINE configs use, small ----
  const int s1[1] = {9}, s2[2] = {7, 15}, s0[1] = {0}, sG[1] = {20};
  if (run_oracle(20, 31, 0, nullptr, true, false)) return 1;
  if (run_oracle(20, 31, 1, s1, true, true)) return 1;
  if (run_oracle(20, 31, 2, s2, false, false)) return 1;
  if (run_oracle(20, 31, 1, s0, true, false)) return 1;  // empty first chunk
  if (run_oracle(20, 31, 1, sG, false, true)) return 1;  // split == n_gaps: empty last chunk (abd.py:615)
  if (run_oracle(2, 1, 0, nullptr, true, false)) return 1;  // smallest legal cohort
  if (run_oracle(200, 9, 2, s2, true, false)) return 1;
  std::puts("sanitize ok");
  return 0;
}
