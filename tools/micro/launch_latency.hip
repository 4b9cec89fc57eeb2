// Launch-latency floor of a synchronous call: kernel(s) that write a tag into mapped host memory, host polls.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/launch_latency.hip -o build/launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

struct Big { double v[256]; };  // a 2 KB kernel argument, like EvalArgs

__global__ void k_small(volatile double* out, double tag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); out[0] = tag; }
}
__global__ void k_big(Big b, volatile double* out, double tag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); out[0] = tag + b.v[7] * 0.0; }
}
__global__ void k_work(double* scratch, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scratch[i] = scratch[i] * 1.0000001 + 1.0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  double* h; hipHostMalloc(&h, 4096, hipHostMallocMapped); memset(h, 0, 4096);
  double* d; hipHostGetDevicePointer((void**)&d, h, 0);
  double* scratch; hipMalloc(&scratch, 1 << 20); hipMemset(scratch, 0, 1 << 20);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  Big big; memset(&big, 0, sizeof big);
  const int iters = 5000;
  volatile double* hv = h;
  for (int variant = 0; variant < 6; ++variant) {
    double t0 = 0;
    for (int it = -500; it < iters; ++it) {
      if (it == 0) t0 = now();
      const double tag = (double)(variant * 100000 + it + 1000);
      switch (variant) {
        case 0: hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d, tag); break;
        case 1: hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, big, d, tag); break;
        case 2: hipLaunchKernelGGL(k_work, dim3(588), dim3(256), 0, st, scratch, 588 * 256);
                hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d, tag); break;
        case 3: hipLaunchKernelGGL(k_work, dim3(588), dim3(256), 0, st, scratch, 588 * 256);
                hipLaunchKernelGGL(k_big, dim3(4), dim3(1024), 0, st, big, d, tag); break;
        case 4: hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, d, tag); hipStreamSynchronize(st); break;
        case 5: hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, 0, d, tag); break;  // null stream
      }
      if (variant != 4) while (hv[0] != tag) __builtin_ia32_pause();
    }
    const double us = (now() - t0) / iters * 1e6;
    const char* names[] = {"1 small kernel + poll", "1 kernel with 2 KB args + poll", "work kernel + small kernel + poll",
                           "work kernel + (4 x 1024, 2 KB args) kernel + poll", "1 small kernel + hipStreamSynchronize",
                           "1 small kernel on the null stream + poll"};
    printf("%-52s %7.2f us\n", names[variant], us);
  }
  return 0;
}
