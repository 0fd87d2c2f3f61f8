// fit_kernels.hip — spline initialisation on the device.
//
// BSpline::FitToData / FitSpline (bspline.hpp:19-37, 246-297) solves the least-squares problem
//   min_C || X C - data ||²,  X(j, si_j + a) = w_a(t_j)  (k non-zeros per row, si = GetSplineIndex(t)),
// with a dense column-pivoted QR on the n×n_ctrl design matrix -- cubic in the trajectory length (the reference's own
// TODO, bspline.hpp:287-289, notes that the system is banded). Here: the normal equations XᵀX C = Xᵀdata are banded
// SPD with half bandwidth k-1, assembled without atomics and solved by a banded Cholesky in LDS.
//   fit_weights_kernel   one thread per sample: spline weights (bspline.hpp:39-72, derivative 0) -> W[n][k]
//   fit_normal_kernel    one thread per entry of [band(n_ctrl, k) | rhs(n_ctrl, 6)]: fixed-order sum over the samples
//                        of the segments that touch the control point (samples are sorted, seg_ptr is a CSR)
//   fit_solve_kernel     one wave: banded Cholesky (tiny ridge; pivots the samples do not determine -- the trajectory end
//                        can have fewer samples than control points, where the reference's QR result is
//                        roundoff-defined -- are decoupled),
//                        forward and backward substitution of the six right-hand sides, all in LDS
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/calico_hip.h"
#include "device_math.hpp"

namespace cal {

__global__ void fit_weights_kernel(int n, int k, const double* __restrict__ stamps, const int* __restrict__ seg,
                                   const double* __restrict__ knots, const double* __restrict__ basis, double* __restrict__ W) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = seg[j], ki = s + k - 1;
  double w[1][kMaxOrder];
  spline_weights<1, 0>(k, knots[ki], knots[ki + 1], basis + size_t(s) * k * k, stamps[j], w);
#pragma unroll
  for (int a = 0; a < kMaxOrder; ++a) if (a < k) W[size_t(j) * k + a] = w[0][a];
}

// entry e of row a: e < k -> N(a, a - e) (lower band), e >= k -> rhs(a, e - k)
__global__ void fit_normal_kernel(int n_ctrl, int n_seg, int k, const double* __restrict__ W, const double* __restrict__ data,
                                  const int* __restrict__ seg_ptr, double* __restrict__ band, double* __restrict__ rhs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = k + 6;
  if (t >= n_ctrl * per) return;
  const int a = t / per, e = t - a * per;
  double acc = 0.0;
  const int s_lo = max(0, a - (k - 1)), s_hi = min(a, n_seg - 1);
  for (int s = s_lo; s <= s_hi; ++s) {
    const int ia = a - s;                       // local index of control point a in segment s
    const int ib = e < k ? ia - e : 0;          // local index of control point a - e
    if (e < k && ib < 0) continue;
    for (int q = seg_ptr[s]; q < seg_ptr[s + 1]; ++q) {
      const double wa = W[size_t(q) * k + ia];
      acc += e < k ? wa * W[size_t(q) * k + ib] : wa * data[size_t(q) * 6 + (e - k)];
    }
  }
  if (e < k) band[size_t(a) * k + e] = acc; else rhs[size_t(a) * 6 + (e - k)] = acc;
}

__global__ __launch_bounds__(64) void fit_solve_kernel(int n, int k, const double* __restrict__ band_g, const double* __restrict__ rhs_g,
                                                       double* __restrict__ ctrl, int* status) {
  extern __shared__ double lds[];
  double* B = lds;                   // [n][k]  B[a][d] = L(a, a - d) after the factorisation
  double* Y = lds + size_t(n) * k;   // [n][6]
  const int lane = threadIdx.x;
  for (int i = lane; i < n * k; i += 64) B[i] = band_g[i];
  for (int i = lane; i < n * 6; i += 64) Y[i] = rhs_g[i];
  __syncthreads();
  double tr = 0.0;
  for (int i = lane; i < n; i += 64) tr += B[i * k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tr += __shfl_xor(tr, off, 64);
  const double mean_diag = tr / n;
  const double ridge = 1e-15 * mean_diag;      // keeps exact zeros away; its effect on a well-posed fit is ~1e-9 relative
  for (int i = lane; i < n; i += 64) B[i * k] += ridge;
  __syncthreads();
  // right-looking banded Cholesky: lane (i, c), 1 <= c <= i < k, owns the update of entry (j+i, j+c)
  int ui = 0, uc = 0;
  {
    int t = lane, i = 1;
    while (i < k && t >= i) { t -= i; ++i; }
    if (i < k) { ui = i; uc = t + 1; }
  }
  bool bad = false;
  for (int j = 0; j < n; ++j) {
    double d = B[j * k];
    bad = bad || !isfinite(d);
    // a control point the samples do not determine (fewer samples than control points at the trajectory end: the
    // reference's pivoted QR returns roundoff there) is decoupled instead of dividing by noise
    if (!(d > 1e-13 * mean_diag)) d = mean_diag;
    const double inv = 1.0 / sqrt(d);
    __syncthreads();
    if (lane == 0) B[j * k] = d * inv;
    if (lane >= 1 && lane < k && j + lane < n) B[(j + lane) * k + lane] *= inv;      // L(j+i, j)
    __syncthreads();
    if (ui > 0 && j + ui < n) B[(j + ui) * k + (ui - uc)] -= B[(j + ui) * k + ui] * B[(j + uc) * k + uc];
    __syncthreads();
  }
  // forward then backward substitution, lane c < 6 per right-hand side
  if (lane < 6) {
    for (int a = 0; a < n; ++a) {
      double s = Y[a * 6 + lane];
      for (int d = 1; d < k && d <= a; ++d) s -= B[a * k + d] * Y[(a - d) * 6 + lane];
      Y[a * 6 + lane] = s / B[a * k];
    }
    for (int a = n - 1; a >= 0; --a) {
      double s = Y[a * 6 + lane];
      for (int d = 1; d < k && a + d < n; ++d) s -= B[(a + d) * k + d] * Y[(a + d) * 6 + lane];
      s /= B[a * k];
      Y[a * 6 + lane] = s;
      ctrl[size_t(a) * 6 + lane] = s;
    }
  }
  if (lane == 0) *status = bad ? 1 : 0;
}

}  // namespace cal

namespace {
template <class T> struct Buf {
  T* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)); }
};
}  // namespace

extern "C" int32_t calico_fit_spline(int32_t device, int32_t order, int32_t n_knots, const double* knots, const double* basis,
                                     int64_t n, const double* stamps, const double* data6, double* ctrl_out) {
  using namespace cal;
  if (order < 2 || order > kMaxOrder || !knots || !basis || !stamps || !data6 || !ctrl_out || n <= 0 || n_knots < 2 * order)
    return CALICO_INVALID_ARGUMENT;
  const int k = order, deg = k - 1, n_ctrl = n_knots - k, n_valid = n_knots - 2 * deg, n_seg = n_valid - 1;
  if (n_seg < 1) return CALICO_INVALID_ARGUMENT;
  // GetSplineIndex (bspline.hpp:138-150): upper_bound(valid_knots, t) - 1; the last valid knot belongs to the last segment
  std::vector<int> seg(static_cast<size_t>(n)), seg_ptr(static_cast<size_t>(n_seg) + 1, 0);
  const double* vk = knots + deg;
  for (int64_t j = 0; j < n; ++j) {
    const double t = stamps[j];
    if (j > 0 && !(t >= stamps[j - 1])) return CALICO_INVALID_ARGUMENT;      // samples must be sorted (trajectory.cpp:24)
    if (!(t >= vk[0]) || !(t <= vk[n_valid - 1])) return CALICO_INVALID_ARGUMENT;
    int s = int(std::upper_bound(vk, vk + n_valid, t) - vk) - 1;
    if (s > n_seg - 1) s = n_seg - 1;
    seg[size_t(j)] = s;
    seg_ptr[size_t(s) + 1] += 1;
  }
  for (int s = 0; s < n_seg; ++s) seg_ptr[size_t(s) + 1] += seg_ptr[size_t(s)];
  const size_t lds = (size_t(n_ctrl) * k + size_t(n_ctrl) * 6) * sizeof(double);
  if (lds > 156 * 1024) return CALICO_UNIMPLEMENTED;   // trajectory too long for the in-LDS banded solve
  int dev_count = 0;
  if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count <= 0 || hipSetDevice(device) != hipSuccess) return CALICO_INTERNAL;
  Buf<double> d_stamps, d_data, d_knots, d_basis, d_W, d_band, d_rhs, d_ctrl;
  Buf<int> d_seg, d_ptr, d_status;
#define TRY(x) do { if ((x) != hipSuccess) return CALICO_INTERNAL; } while (0)
  TRY(d_stamps.alloc(size_t(n))); TRY(d_data.alloc(size_t(n) * 6)); TRY(d_knots.alloc(size_t(n_knots)));
  TRY(d_basis.alloc(size_t(n_seg) * k * k)); TRY(d_W.alloc(size_t(n) * k)); TRY(d_band.alloc(size_t(n_ctrl) * k));
  TRY(d_rhs.alloc(size_t(n_ctrl) * 6)); TRY(d_ctrl.alloc(size_t(n_ctrl) * 6)); TRY(d_seg.alloc(size_t(n)));
  TRY(d_ptr.alloc(size_t(n_seg) + 1)); TRY(d_status.alloc(1));
  TRY(hipMemcpy(d_stamps.p, stamps, size_t(n) * sizeof(double), hipMemcpyHostToDevice));
  TRY(hipMemcpy(d_data.p, data6, size_t(n) * 6 * sizeof(double), hipMemcpyHostToDevice));
  TRY(hipMemcpy(d_knots.p, knots, size_t(n_knots) * sizeof(double), hipMemcpyHostToDevice));
  TRY(hipMemcpy(d_basis.p, basis, size_t(n_seg) * k * k * sizeof(double), hipMemcpyHostToDevice));
  TRY(hipMemcpy(d_seg.p, seg.data(), size_t(n) * sizeof(int), hipMemcpyHostToDevice));
  TRY(hipMemcpy(d_ptr.p, seg_ptr.data(), (size_t(n_seg) + 1) * sizeof(int), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fit_weights_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, int(n), k, d_stamps.p, d_seg.p, d_knots.p,
                     d_basis.p, d_W.p);
  const int n_entries = n_ctrl * (k + 6);
  hipLaunchKernelGGL(fit_normal_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, 0, n_ctrl, n_seg, k, d_W.p, d_data.p, d_ptr.p,
                     d_band.p, d_rhs.p);
  TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  hipLaunchKernelGGL(fit_solve_kernel, dim3(1), dim3(64), lds, 0, n_ctrl, k, d_band.p, d_rhs.p, d_ctrl.p, d_status.p);
  int status = 0;
  TRY(hipMemcpy(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost));
  TRY(hipMemcpy(ctrl_out, d_ctrl.p, size_t(n_ctrl) * 6 * sizeof(double), hipMemcpyDeviceToHost));
#undef TRY
  return status == 0 ? CALICO_OK : CALICO_INTERNAL;
}
