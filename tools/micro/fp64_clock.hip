// Sustained fp64 FMA rate / effective shader clock under an all-SIMD fp64 load.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = s;
    out[blockIdx.x * 4 + 1] = (double)(t1 - t0);
    out[blockIdx.x * 4 + 2] = (double)(r1 - r0);
  }
}
int main() {
  const int blocks = 1024, iters = 20000;
  double* d; hipMalloc(&d, blocks * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int q = 0; q < 20; ++q) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999999, 1e-7);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    double h[blocks * 4]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0; for (int b = 0; b < blocks; ++b) { cyc += h[b*4+1]; real += h[b*4+2]; }
    cyc /= blocks; real /= blocks;  // s_memrealtime ticks at 100 MHz
    double wave_instr_per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
    const double ghz = cyc / real * 0.1;  // memtime ticks per memrealtime tick (100 MHz)
    printf("kernel %.3f ms | %.1f TFLOP/s fp64 | s_memtime/s_memrealtime -> %.3f GHz | memtime ticks per wave-fma per SIMD %.2f | implied clock at 4 cyc/fma %.2f GHz\n",
           ms, (double)blocks * 256 * iters * 8 * 2 / (ms * 1e-3) / 1e12, ghz, cyc / wave_instr_per_simd,
           wave_instr_per_simd * 4 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
