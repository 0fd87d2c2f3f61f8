// stage_b_rate.hip -- one work item's [J r]^T [J r] (stage_b_mfma of eval_kernels.hip, NT = 4: ten 16x16 tiles, 60 staged rows =
// 15 k-steps = 150 v_mfma_f64_16x16x4_f64) alone on a CU: clocks for the whole call, so that what the evaluation's dev-timing
// dump shows for it (~17k clocks) can be split into products (150 x 64 = 9.6k), the call's prologue and the tile stores.
#include "../../calico_amd/csrc/eval_kernels.hip"
#include <cstdio>
namespace cal {
__global__ __launch_bounds__(64) void stage_b_bench(double* out, long long* ticks, int pad, int nrows, int n1, int reps) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 64 * pad; i += 64) lds[i] = (i % pad < nrows && i / pad < n1) ? 1e-3 * double(i % 97) : 0.0;
  __syncthreads();
  for (int r = 0; r < reps; ++r) {
    const long long t0 = __builtin_readcyclecounter();
    stage_b_dispatch<0>(lds, pad, nrows, n1, out);
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) ticks[r] = t1 - t0;
  }
}
}
int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 64 * 64 * 8); hipMalloc(&ticks, 64 * 8);
  const int pad = 61, nrows = 60, reps = 8;
  for (int n1 : {53, 56, 35}) {
    hipLaunchKernelGGL(cal::stage_b_bench, dim3(1), dim3(64), 64 * pad * 8, 0, out, ticks, pad, nrows, n1, reps);
    long long h[64];
    hipMemcpy(h, ticks, reps * 8, hipMemcpyDeviceToHost);
    const int NT = (n1 + 15) / 16, tiles = NT * (NT + 1) / 2, ks = (nrows + 3) / 4;
    std::printf("n1 %d (%d tiles x %d k-steps = %d products = %d clocks at 64): calls took", n1, tiles, ks, tiles * ks, tiles * ks * 64);
    for (int r = 0; r < reps; ++r) std::printf(" %lld", h[r]);
    std::printf(" clocks\n");
  }
  return 0;
}
