// Accuracy of v_rcp_f64 and of 1 / 2 Newton steps on it, and of exp_reduced, measured on the device.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../abdpymc_amd/csrc/abd_device.hpp"

__global__ void k(const double* x, const double* arg, double* r0, double* r1, double* r2, double* ex, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  r = fma(fma(-d, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-d, r, 1.0), r, r);
  r2[i] = r;
  ex[i] = exp2_reduced(arg[i]);  // the kernel's 2^t form
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n), e(n), arg(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (s >> 11) * (1.0 / 9007199254740992.0);
    x[i] = std::exp(u * 40.0);  // [1, e^40]
    arg[i] = u * 2042.0 - 1021.0;   // t in [-1021, 1021]
  }
  double *dx, *d0, *d1, *d2, *de, *da;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&de, n * 8); hipMalloc(&da, n * 8);
  hipMemcpy(da, arg.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, d0, d1, d2, de, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, me = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / (long double)x[i];
    m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
    m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
    m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
    long double te = exp2l((long double)arg[i]);
    me = fmax(me, (double)fabsl(((long double)e[i] - te) / te));
  }
  printf("v_rcp_f64 max rel err %.3e | +1 Newton %.3e | +2 Newton %.3e | exp2_reduced %.3e\n", m0, m1, m2, me);
  return 0;
}
