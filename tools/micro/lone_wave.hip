// What one wave can issue when it has a SIMD to itself: cycles per v_fma_f64 (and per v_and_b32) for 1, 2 and 4 waves per
// SIMD and 1, 2, 4, 8 independent dependency chains per wave (s_memtime around a long unrolled loop).
// hipcc -O3 --offload-arch=gfx950 tools/micro/lone_wave.hip -o /tmp/lone_wave && /tmp/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 2000
template <int ILP, bool F64>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, double a, double b) {
  double x[8];
  int y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + 1.5 + i; y[i] = threadIdx.x + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / ILP; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        else asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(y[i]));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; int sy = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { s += x[i]; sy += y[i]; }
  if (s == 1.2345 && sy == 77) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int ILP, bool F64>
void run(int waves_per_simd) {
  unsigned long long* d;
  (void)hipMalloc(&d, 16);
  const int blocks = waves_per_simd > 4 ? 512 : 256, threads = waves_per_simd > 4 ? 1024 : 256 * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<ILP, F64><<<blocks, threads>>>(d, 0.999, 1e-3);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) k<ILP, F64><<<blocks, threads>>>(d, 0.999, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double ns_per_inst_simd = ms * 1e6 / 10 / (ITERS * 8.0 * waves_per_simd);
  printf("%s waves/SIMD %d  ILP %d : s_memtime %.2f ticks per instruction of one wave; wall %.2f ns per instruction of the SIMD = %.2f cycles at 2.1 GHz\n",
         F64 ? "v_fma_f64" : "v_and_b32", waves_per_simd, ILP, (double)h / (ITERS * 8.0), ns_per_inst_simd, ns_per_inst_simd * 2.1);
  (void)hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<1, true>(w); run<2, true>(w); run<4, true>(w); run<8, true>(w); }
  for (int w : {1, 2, 4, 8}) { run<1, false>(w); run<2, false>(w); run<8, false>(w); }
  return 0;
}
