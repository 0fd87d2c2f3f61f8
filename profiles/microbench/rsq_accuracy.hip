// Accuracy of v_rsq_f64 and of one / two Newton steps on top of it (relative error vs. 1/sqrt in long double on host).
//   hipcc --offload-arch=gfx950 -O3 rsq_accuracy.hip -o rsq_accuracy && ./rsq_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r = __builtin_amdgcn_rsq(d);
  r0[i] = r;
  r = r * (1.5 - 0.5 * d * r * r);
  r1[i] = r;
  r = r * (1.5 - 0.5 * d * r * r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (s >> 11) * 0x1.0p-53, int(s % 80) - 40); }
  double *dx, *d0, *d1, *d2;
  (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d0, n * 8); (void)hipMalloc(&d1, n * 8); (void)hipMalloc(&d2, n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  std::vector<double> r0(n), r1(n), r2(n);
  (void)hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  long double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / sqrtl((long double)x[i]);
    e0 = fmaxl(e0, fabsl(r0[i] - t) / t); e1 = fmaxl(e1, fabsl(r1[i] - t) / t); e2 = fmaxl(e2, fabsl(r2[i] - t) / t);
  }
  printf("max relative error: v_rsq_f64 %.3Le (2^%.1Lf)  +1 Newton %.3Le  +2 Newton %.3Le   (eps = 2.22e-16)\n", e0, log2l(e0), e1, e2);
  return 0;
}
