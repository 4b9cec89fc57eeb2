// What does a SIMD issue with 1, 2, 3 or 4 waves on it?  The same kernel -- a loop of 8 independent fp64 FMA chains per lane --
// as long kernels, as short kernels back to back and as short kernels each waited for; "implied clock" = (wave-instructions per
// SIMD x 4 cycles) / time, i.e. the clock the part would have to run at if it issued an FMA every 4 cycles.  Result (round 4,
// profiles/r04/b_lone_wave_issue.txt): 1.26 / 1.68 / 1.88 / 1.97 GHz with 1 / 2 / 3 / 4 waves per SIMD = 0.64, 0.85, 0.955, 1.0 of
// the rate -- exactly 1 - 0.36^n: a wave is ready to issue 64 % of the time (the loop's branch and the in-order issue cost a lone
// wave what other waves would fill), independently of the others.  Not a clock effect.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  const double s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (s == 12345.678) out[blockIdx.x] = s;
}
static void run(const char* what, int blocks, int iters, int launches, bool wait_each, double* d) {
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < launches; ++q) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999999, 1e-7);
      if (wait_each) hipDeviceSynchronize();
    }
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / launches;
    const double inst_per_simd = (double)blocks * 4 * iters * 8 / 1024.0;  // 256 CUs x 4 SIMDs
    if (rep == 2)
      std::printf("%-58s %8.1f us per launch, implied clock %.2f GHz (incl. launch overhead)\n", what, us, inst_per_simd * 4 / (us * 1e3));
  }
}
int main() {
  double* d;
  hipMalloc(&d, 4096 * 8);
  run("4 waves per SIMD, long kernels (1.3 ms)", 1024, 20000, 10, false, d);
  run("1 wave per SIMD, long kernels", 256, 80000, 10, false, d);
  run("2 waves per SIMD, long kernels", 512, 40000, 10, false, d);
  run("3 waves per SIMD, long kernels", 768, 26667, 10, false, d);
  run("4 waves per SIMD, 15 us kernels back to back", 1024, 220, 400, false, d);
  run("1 wave per SIMD, 15 us kernels back to back", 256, 880, 400, false, d);
  run("1 wave per SIMD, 15 us kernels, each waited for", 256, 880, 400, true, d);
  run("4 waves per SIMD, 15 us kernels, each waited for", 1024, 220, 400, true, d);
  run("1 wave per SIMD, 100 us kernels, each waited for", 256, 5900, 100, true, d);
  return 0;
}
