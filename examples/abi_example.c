/*
 * The C ABI from plain C (C99): build a small dense cohort, evaluate logp + gradient, run the compound sampler,
 * read the posterior means back.  What a non-Python host (or a cffi / cgo / JNI stub) would do.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include examples/abi_example.c -L abdpymc_amd -labd_hip \
 *       -Wl,-rpath,$PWD/abdpymc_amd -lm -o build/abi_example
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "abd_hip.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ != ABD_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, abd_last_error());       \
      return 1;                                                              \
    }                                                                        \
  } while (0)

int main(void) {
  enum { N = 96, G = 24, C = 2, K = N * G };
  static int32_t gap[K], ind[K];
  static double x[K], y_s[K], y_n[K];
  static int8_t vacs[K], pcr[K], i_raw[K], waner[N];
  unsigned s = 12345u;
  for (int j = 0; j < N; ++j)
    for (int g = 0; g < G; ++g) {
      const int k = j * G + g;
      s = s * 1664525u + 1013904223u;
      gap[k] = g;
      ind[k] = j;
      x[k] = 2.0 * ((s >> 8) % 3);
      /* one infection at gap 8 for every third individual: OD rises from ~0 to ~1.6 afterwards */
      const int infected = (j % 3 == 0) && g >= 8;
      y_n[k] = (infected && x[k] < 3.0 ? 1.5 : 0.03) + 0.01 * ((s >> 20) % 7);
      y_s[k] = y_n[k];
      vacs[k] = 0;
      pcr[k] = 0;
      i_raw[k] = 0;
    }
  for (int j = 0; j < N; ++j) waner[j] = 1;

  abd_desc d;
  d.n_gaps = G;
  d.n_inds = N;
  d.n_splits = 0;
  d.splits[0] = d.splits[1] = 0;
  d.storage = ABD_STORE_F64;
  d.n_chain_slots = C;
  d.device = -1;
  d.s.n_obs = K; d.s.idx_gap = gap; d.s.idx_ind = ind; d.s.log_dilution = x; d.s.od = y_s;
  d.n = d.s;
  d.n.od = y_n;
  d.vacs = vacs;
  d.pcrpos = pcr;

  abd_ctx* ctx = NULL;
  CHECK(abd_create(&d, &ctx));
  char name[256];
  CHECK(abd_device_name(ctx, name, (int32_t)sizeof name));
  printf("%s on %s, dense panels: %d\n", abd_version(), name, abd_is_dense(ctx));

  /* i_raw is (G, N) as PyMC holds it; all zeros here */
  for (int c = 0; c < C; ++c) CHECK(abd_set_discrete(ctx, c, i_raw, waner));

  double theta[C * ABD_N_THETA];
  for (int c = 0; c < C; ++c) {
    double* t = theta + c * ABD_N_THETA;
    const double init[ABD_N_THETA] = {-3.1, 0.69, 0.0, 2.3, -2.0, 0.69, 2.3, 0.0, 0.0, 0.0, -2.0, -1.0, 2.0, 0.0, -1.0, 2.0, 0.0};
    for (int k = 0; k < ABD_N_THETA; ++k) t[k] = init[k] + 0.01 * c;
  }
  const int32_t chains[C] = {0, 1};
  double logp[C], grad[C * ABD_N_THETA];
  CHECK(abd_logp_dlogp_batch(ctx, C, chains, theta, logp, grad));
  printf("logp = %.6f, %.6f; d logp / d it_n_d = %.6f\n", logp[0], logp[1], grad[12]);
  if (!isfinite(logp[0]) || !isfinite(grad[12])) return 1;

  abd_sampler_opts o;
  o.tune = 150;
  o.seed = 7;
  o.target_accept = 0.8;
  o.max_treedepth = 10;
  o.gibbs = 1;
  o.accumulate = 1;
  o.chain_offset = 0;
  o.dense_metric = 1;
  o.reserved = 0;
  abd_sampler* smp = NULL;
  CHECK(abd_sampler_create(ctx, C, chains, theta, &o, &smp));
  enum { DRAWS = 150 };
  static double draws[C * DRAWS * ABD_N_THETA], stats[C * DRAWS * ABD_N_STATS];
  CHECK(abd_sampler_run(smp, o.tune, NULL, NULL));
  CHECK(abd_sampler_run(smp, DRAWS, draws, stats));
  static double i_mean[K];
  int64_t n_draws = 0;
  CHECK(abd_sampler_means(smp, 0, i_mean, NULL, NULL, &n_draws));
  double found = 0.0, spurious = 0.0;
  for (int j = 0; j < N; ++j) {
    double any = 0.0; /* posterior mass of "infected somewhere in gaps 5..8" */
    for (int g = 5; g <= 8; ++g) any += i_mean[g * N + j];
    if (j % 3 == 0) found += any; else spurious += any;
  }
  found /= N / 3;
  spurious /= N - N / 3;
  double d_n = 0.0;
  for (int k = 0; k < DRAWS; ++k) d_n += draws[k * ABD_N_THETA + 12];
  printf("%lld draws: it_n_d = %.3f, infection found for %.0f%% of the infected, %.1f%% of the others\n", (long long)n_draws,
         d_n / DRAWS, 100.0 * found, 100.0 * spurious);
  abd_sampler_destroy(smp);
  CHECK(abd_destroy(ctx));
  return (found > 0.8 && spurious < 0.1) ? 0 : 2;
}
