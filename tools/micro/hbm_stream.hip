// Achievable HBM bandwidth on this part: copy and triad over buffers far larger than the 256 MiB Infinity Cache.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_stream.hip -o build/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const double2* __restrict__ a, double* __restrict__ out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i].x + a[i].y;
  if (s == 1.2345e300) out[0] = s;
}
__global__ void k_triad(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double2 x = a[i], y = b[i];
    c[i] = make_double2(x.x + 3.0 * y.x, x.y + 3.0 * y.y);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t bytes = (size_t)2 << 30;  // 2 GiB per buffer
  const size_t n = bytes / sizeof(double2);
  double2 *a, *b, *c;
  double* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 16, block = 256;
  for (int which = 0; which < 4; ++which) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, 0));
      if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, a, out, n);
      if (which == 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(block), 0, 0, a, b, n);
      if (which == 2) hipLaunchKernelGGL(k_triad, dim3(grid), dim3(block), 0, 0, a, b, c, n);
      if (which == 3) CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const double moved = (which == 0 ? 1.0 : which == 2 ? 3.0 : 2.0) * bytes;
    const char* names[] = {"read  (2 GiB)", "copy  (2 GiB -> 2 GiB)", "triad (2 x 2 GiB -> 2 GiB)", "hipMemcpy DtoD (2 GiB)"};
    printf("%-28s %7.3f ms  %7.1f GB/s\n", names[which], best, moved / best / 1e6);
  }
  return 0;
}
