// Relative issue cost of the fp64-side VALU instructions the dense kernel uses, against v_fma_f64.
// Each kernel runs a long loop of 8 independent copies of one instruction (inline asm), 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#define ITERS 4000
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

#define KERNEL(NAME, ASM)                                                                       \
  __global__ __launch_bounds__(256) void NAME(double* out, double a, double b) {                \
    double x0 = threadIdx.x + 1.5, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    int i0 = threadIdx.x & 3;                                                                   \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      asm volatile(ASM(0) : "+v"(x0) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x1) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x2) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x3) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x4) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x5) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x6) : "v"(a), "v"(b), "v"(i0));                                \
      asm volatile(ASM(0) : "+v"(x7) : "v"(a), "v"(b), "v"(i0));                                \
    }                                                                                           \
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                \
  }
#define A_FMA(n) "v_fma_f64 %0, %0, %1, %2"
#define A_MUL(n) "v_mul_f64 %0, %0, %1"
#define A_ADD(n) "v_add_f64 %0, %0, %2"
#define A_MIN(n) "v_min_f64 %0, %0, %1"
#define A_RCP(n) "v_rcp_f64 %0, %0"
#define A_RND(n) "v_rndne_f64 %0, %0"
#define A_LDEXP(n) "v_ldexp_f64 %0, %0, %3"
#define A_SHR(n) "v_lshrrev_b64 %0, 1, %0"
#define A_MOV(n) "v_mov_b64 %0, %1"
KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_min, A_MIN) KERNEL(k_rcp, A_RCP)
KERNEL(k_rnd, A_RND) KERNEL(k_ldexp, A_LDEXP) KERNEL(k_shr, A_SHR) KERNEL(k_mov, A_MOV)

// 32-bit destination forms
#define KERNEL32(NAME, ASM)                                                                     \
  __global__ __launch_bounds__(256) void NAME(double* out, double a, double b) {                \
    double x = threadIdx.x + 1.5;                                                               \
    int y0 = 1, y1 = 2, y2 = 3, y3 = 4, y4 = 5, y5 = 6, y6 = 7, y7 = 8;                         \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      asm volatile(ASM : "+v"(y0) : "v"(x)); asm volatile(ASM : "+v"(y1) : "v"(x));             \
      asm volatile(ASM : "+v"(y2) : "v"(x)); asm volatile(ASM : "+v"(y3) : "v"(x));             \
      asm volatile(ASM : "+v"(y4) : "v"(x)); asm volatile(ASM : "+v"(y5) : "v"(x));             \
      asm volatile(ASM : "+v"(y6) : "v"(x)); asm volatile(ASM : "+v"(y7) : "v"(x));             \
    }                                                                                           \
    out[blockIdx.x * 256 + threadIdx.x] = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + a + b;        \
  }
KERNEL32(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1")
KERNEL32(k_and_b32, "v_and_b32 %0, 1, %0")
__global__ __launch_bounds__(256) void k_cvt_f64_u32(double* out, double a, double b) {
  double x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0, x6 = 0, x7 = 0;
  unsigned u = threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x0) : "v"(u)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x1) : "v"(u));
    asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x2) : "v"(u)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x3) : "v"(u));
    asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x4) : "v"(u)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x5) : "v"(u));
    asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x6) : "v"(u)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x7) : "v"(u));
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + a + b;
}

template <typename K>
double run(K k, double* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 0.999999, 1e-7);
  hipEventRecord(e0);
  for (int q = 0; q < 5; ++q) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 0.999999, 1e-7);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  double* d; hipMalloc(&d, 1024 * 256 * 8);
  const double base = run(k_fma, d);
  struct { const char* n; double ms; } r[] = {
    {"v_fma_f64", base}, {"v_mul_f64", run(k_mul, d)}, {"v_add_f64", run(k_add, d)}, {"v_min_f64", run(k_min, d)},
    {"v_rcp_f64", run(k_rcp, d)}, {"v_rndne_f64", run(k_rnd, d)}, {"v_ldexp_f64", run(k_ldexp, d)},
    {"v_lshrrev_b64", run(k_shr, d)}, {"v_mov_b64", run(k_mov, d)}, {"v_cvt_i32_f64", run(k_cvt_i32_f64, d)},
    {"v_cvt_f64_u32", run(k_cvt_f64_u32, d)}, {"v_and_b32", run(k_and_b32, d)}, {"v_fma_f64 (again)", run(k_fma, d)}};
  for (auto& x : r) printf("%-20s %8.3f ms   %.2f x v_fma_f64\n", x.n, x.ms, x.ms / base);
  return 0;
}
