// Micro-benchmark: sums across lanes without the LDS crossbar. __shfl_xor compiles to ds_bpermute_b32 (two per double);
// on a dependent chain every step pays the LDS round trip. Inside a row of 16 lanes the DPP modifiers exchange lanes in
// two v_mov_b32_dpp; v_permlane32_swap / v_permlane16_swap (gfx950) exchange the halves of a wave and the rows of a half.
// Checks the results against the shuffle versions and times N dependent reductions of each kind in one wave.
//   hipcc --offload-arch=gfx950 -O3 lane_sums.hip -o /tmp/lane_sums && /tmp/lane_sums
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
__device__ __forceinline__ double other_half(double v) {     // lane i receives lane i ^ 32
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const bool up = (threadIdx.x & 32) != 0;
  return __hiloint2double(up ? b[0] : b[1], up ? a[0] : a[1]);
}
__device__ __forceinline__ double other_row(double v) {      // lane i receives lane i ^ 16
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const bool odd = (threadIdx.x & 16) != 0;
  return __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
}
__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  v += other_half(v);
  v += other_row(v);
  return v;
}
__device__ __forceinline__ double wave_sum_shfl(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double row16_sum_shfl(double v) {
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
  return v;
}

__global__ __launch_bounds__(64) void check(const double* in, double* out, long long* clocks, int n) {
  const int lane = threadIdx.x;
  const double v = in[lane];
  out[lane] = row16_sum(v); out[64 + lane] = row16_sum_shfl(v);
  out[128 + lane] = wave_sum(v); out[192 + lane] = wave_sum_shfl(v);
  out[256 + lane] = other_half(v); out[320 + lane] = __shfl_xor(v, 32, 64);
  out[384 + lane] = other_row(v); out[448 + lane] = __shfl_xor(v, 16, 64);
  double x = v;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) x = row16_sum(x) * 0.0625;
  long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) x = row16_sum_shfl(x) * 0.0625;
  long long t2 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) x = wave_sum(x) * 0.015625;
  long long t3 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) x = wave_sum_shfl(x) * 0.015625;
  long long t4 = __builtin_readcyclecounter();
  out[512 + lane] = x;
  if (lane == 0) { clocks[0] = t1 - t0; clocks[1] = t2 - t1; clocks[2] = t3 - t2; clocks[3] = t4 - t3; }
}

int main() {
  double h_in[64], h_out[576];
  for (int i = 0; i < 64; ++i) h_in[i] = 1.0 + 0.37 * i + 1e-3 * i * i;
  double *d_in, *d_out; long long* d_c; long long h_c[4];
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_c, sizeof(h_c));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  const int n = 1000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d_in, d_out, d_c, n);
  hipDeviceSynchronize();
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost); hipMemcpy(h_c, d_c, sizeof(h_c), hipMemcpyDeviceToHost);
  double e16 = 0, e64 = 0; int bad = 0;
  for (int i = 0; i < 64; ++i) {
    e16 = fmax(e16, fabs(h_out[i] - h_out[64 + i]) / fabs(h_out[64 + i]));
    e64 = fmax(e64, fabs(h_out[128 + i] - h_out[192 + i]) / fabs(h_out[192 + i]));
    bad += h_out[256 + i] != h_out[320 + i]; bad += h_out[384 + i] != h_out[448 + i];
  }
  printf("row16_sum vs shuffle: max rel diff %.2e; wave_sum vs shuffle: %.2e; exchange mismatches: %d\n", e16, e64, bad);
  printf("clocks per dependent reduction (one wave): row16 dpp %.0f  row16 shuffle %.0f | wave dpp+permlane %.0f  wave shuffle %.0f\n",
         double(h_c[0]) / n, double(h_c[1]) / n, double(h_c[2]) / n, double(h_c[3]) / n);
  return bad != 0 || e16 > 1e-14 || e64 > 1e-14;
}
