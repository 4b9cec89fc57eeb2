// host time of one hipLaunchKernelGGL as a function of the size of the by-value kernel argument (and of dynamic LDS)
// hipcc -O2 --offload-arch=gfx950 tools/micro/launch_args.hip -o build/launch_args && build/launch_args
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N>
struct Args {
  double v[N];
};
template <int N>
__global__ void k(Args<N> a, double* out) {
  if (a.v[0] == 12345.0) out[0] = a.v[N - 1];
}
template <int N>
void run(hipStream_t st, double* d, size_t lds) {
  Args<N> a{};
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), lds, st, a, d);
  hipStreamSynchronize(st);
  const int reps = 2000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), lds, st, a, d);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(st);
  auto t2 = std::chrono::steady_clock::now();
  printf("args %5zu B, dynamic LDS %6zu B: %.2f us of host time per launch (%.2f us per launch incl. drain)\n", sizeof(a), lds,
         std::chrono::duration<double>(t1 - t0).count() / reps * 1e6, std::chrono::duration<double>(t2 - t0).count() / reps * 1e6);
}
int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  double* d;
  hipMalloc(&d, 64);
  run<4>(st, d, 0);
  run<32>(st, d, 0);
  run<96>(st, d, 0);
  run<256>(st, d, 0);
  run<256>(st, d, 2048);
  run<480>(st, d, 0);
  run<4>(st, d, 2048);
  return 0;
}
