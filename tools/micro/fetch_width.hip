// Calibration of rocprofv3's FETCH_SIZE by access width (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Each kernel streams 1 GiB exactly once with one load
// width per lane -- 16 B (dwordx4: the gap rows of abd_dense_kernel<R,CB,GRAD,false>), 8 B (the od rows of the fp64
// split panels), 4 B (the od rows of the fp32 split panels), 1 B (the dictionary codes) -- as buffer-style
// coalesced rows (lane = consecutive element).  Run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- ./build/fetch_width
// and divide 1 GiB by FETCH_SIZE (KiB) * 1024 per kernel: that is the factor for the width.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_width.hip -o build/fetch_width
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <class T>
__device__ double as_double(T v);
template <> __device__ double as_double(double2 v) { return v.x + v.y; }
template <> __device__ double as_double(double v) { return v; }
template <> __device__ double as_double(float v) { return v; }
template <> __device__ double as_double(uint8_t v) { return v; }

template <class T>
__global__ void k_read(const T* __restrict__ a, double* __restrict__ out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += as_double<T>(a[i]);
  if (s == 1.2345e300) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class T>
int run(const char* name, void* buf, double* out, size_t bytes) {
  const size_t n = bytes / sizeof(T);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_read<T>, dim3(256 * 16), dim3(256), 0, 0, (const T*)buf, out, n);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-10s %zu bytes per launch, %7.3f ms, %7.1f GB/s\n", name, bytes, best, bytes / best / 1e6);
  return 0;
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: four times the Infinity Cache
  void* a;
  double* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes));
  if (run<double2>("16B/lane", a, out, bytes)) return 1;
  if (run<double>("8B/lane", a, out, bytes)) return 1;
  if (run<float>("4B/lane", a, out, bytes)) return 1;
  if (run<uint8_t>("1B/lane", a, out, bytes)) return 1;
  return 0;
}
