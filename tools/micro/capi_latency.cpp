// Synchronous-call latency through the C ABI without Python: a 64 x 4 dense cohort, 4 chains.
// build: g++ -O2 -I include tools/micro/capi_latency.cpp -L abdpymc_amd -labd_hip -Wl,-rpath,$PWD/abdpymc_amd -o build/capi_latency
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "abd_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64, G = argc > 2 ? atoi(argv[2]) : 4, C = 4;
  std::vector<int32_t> gi, ji;
  std::vector<double> x, y;
  for (int j = 0; j < N; ++j)
    for (int g = 0; g < G; ++g) { gi.push_back(g); ji.push_back(j); x.push_back(2.0); y.push_back(0.5 + 0.01 * ((j * 7 + g) % 13)); }
  std::vector<int8_t> vacs((size_t)N * G, 0), pcr((size_t)N * G, 0), iraw((size_t)N * G, 0), w(N, 1);
  abd_desc d{};
  d.n_gaps = G; d.n_inds = N; d.n_splits = 0; d.storage = ABD_STORE_F64; d.n_chain_slots = C; d.device = -1;
  d.s = {(int64_t)gi.size(), gi.data(), ji.data(), x.data(), y.data()};
  d.n = d.s;
  d.vacs = vacs.data(); d.pcrpos = pcr.data();
  abd_ctx* ctx = nullptr;
  if (abd_create(&d, &ctx)) { printf("create: %s\n", abd_last_error()); return 1; }
  for (int c = 0; c < C; ++c) abd_set_discrete(ctx, c, iraw.data(), w.data());
  double theta[4 * 17] = {0};
  for (int c = 0; c < C; ++c) { double* t = theta + 17 * c; t[0] = -3; t[1] = 0.7; t[3] = 2; t[4] = -2; t[5] = 0.7; t[6] = 2; t[10] = -2; t[11] = -1; t[12] = 2; t[14] = -1; t[15] = 2; }
  int32_t ids[4] = {0, 1, 2, 3};
  double lp[4], gr[4 * 17];
  for (int n : {1, 4}) {
    double t0 = 0;
    for (int it = -500; it < 5000; ++it) {
      if (it == 0) t0 = now();
      theta[2] = 1e-6 * it;
      if (abd_logp_dlogp_batch(ctx, n, ids, theta, lp, gr)) { printf("eval: %s\n", abd_last_error()); return 1; }
    }
    printf("N=%d G=%d, %d chain(s): %.2f us per abd_logp_dlogp_batch (lp %.6f)\n", N, G, n, (now() - t0) / 5000 * 1e6, lp[0]);
  }
  abd_destroy(ctx);
  return 0;
}
