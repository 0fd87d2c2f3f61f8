// Micro-benchmark: cost (shader clocks) of the in-wave 16-column panel factorisation the linear-solve kernels are built
// on, and of the pieces of its dependency chain, one wave alone on its SIMD.
//   hipcc --offload-arch=gfx950 -O3 -I ../../calico_amd/csrc panel_latency.hip -o /tmp/panel_latency && /tmp/panel_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "solve_dev.hpp"

using namespace cal;
#define PIN(v) asm volatile("" : "+v"(v) : : "memory")

// variant of panel_factor<1> with the trailing updates (LDS broadcast) switched off: the chain alone
template <bool TRAIL>
__device__ __forceinline__ void panel_variant(double* A, int LD, double* bcast, int lane, double* pmin) {
  double av[16];
  const double* src = A + lane * LD;
#pragma unroll
  for (int c = 0; c < 16; ++c) av[c] = src[c];
  double lprev = 0.0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    double lb[16];
    if (TRAIL && jj > 0) {
      const double* br = bcast + ((jj - 1) & 1) * 64;
#pragma unroll
      for (int c = jj + 2; c < 16; ++c) lb[c] = br[c];
    }
    const double pv = readlane_f64(av[jj], jj);
    *pmin = fmin(*pmin, pv);
    const double rs = rsqrt_nr(pv);
    const double l = av[jj] * rs;
    av[jj] = l;
    if (TRAIL) (bcast + (jj & 1) * 64)[lane] = l;
    if (jj + 1 < 16) av[jj + 1] -= l * readlane_f64(l, jj + 1);
    if (jj + 2 < 16) av[jj + 2] -= l * readlane_f64(l, jj + 2);
    if (TRAIL && jj > 0) {
#pragma unroll
      for (int c = jj + 2; c < 16; ++c) av[c] -= lprev * lb[c];
    }
    lprev = l;
    __builtin_amdgcn_sched_barrier(0);
  }
  double* dst = A + lane * LD;
#pragma unroll
  for (int c = 0; c < 16; ++c) dst[c] = av[c];
}

__global__ void bench(double* out, long long* cyc, double seed) {
  __shared__ double A[64 * 33], bcast[128], dinv[80];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long t0, t1;
  int k = 0;
  auto init = [&]() {
    for (int c = 0; c < 32; ++c) A[lane * 33 + c] = (lane == c ? 40.0 : 0.0) + 1.0 / (1 + lane + c) + seed;
    __syncthreads();
  };
  double pmin = 1.0, x = seed + 1.0 + lane * 1e-3;
  // (0) panel_factor<1,false>, (1) panel_factor<1,true>, (2) variant with trailing updates, (3) chain only
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      init();
      PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
      if (wave == 0) {
        if (mode == 0) panel_factor<1, false, false>(A, 33, dinv, bcast, 0, 63, 16, lane, &pmin);
        else if (mode == 1) panel_factor<1, true>(A, 33, dinv, bcast, 0, 63, 16, lane, &pmin);
        else if (mode == 2) panel_variant<true>(A, 33, bcast, lane, &pmin);
        else panel_variant<false>(A, 33, bcast, lane, &pmin);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
      if (threadIdx.x == 0) cyc[k] = t1 - t0;
      ++k;
      __syncthreads();
    }
  }
  // (12) 256 dependent v_fma_f64, (13) 256 dependent v_mul_f64, (14) 64 dependent rsq+add, (15) 64 x [readlane pair -> fma],
  // (16) 64 dependent rsqrt_nr
  const double y = 1.0000001, z = 0.5;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = __builtin_fma(x, y, z);
  PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
  if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  x = x * 1e-30 + 1.0;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = x * y;
  PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
  if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
  PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
  if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { const double s = readlane_f64(x, i & 63); x = __builtin_fma(x, 0.999, s * 1e-9); }
  PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
  if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  x = x * 1e-30 + 2.0;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) x = rsqrt_nr(x) + 1.5;
  PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
  if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // (17) 64 x [readlane pair with SGPR operand straight into the fma]
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { const double s = readlane_f64(x, i & 63); x = __builtin_fma(x, 0.5, s); x = x * 1e-9 + 1.0; }
  PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
  if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  out[threadIdx.x] = x + pmin + A[lane];
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 512 * sizeof(double)); hipMalloc(&cyc, 64 * sizeof(long long));
  for (int threads : {64, 512}) {
    hipMemset(cyc, 0, 64 * sizeof(long long));
    hipLaunchKernelGGL(bench, dim3(1), dim3(threads), 0, 0, out, cyc, 0.25);
    hipLaunchKernelGGL(bench, dim3(1), dim3(threads), 0, 0, out, cyc, 0.25);
    hipDeviceSynchronize();
    std::vector<long long> h(64);
    hipMemcpy(h.data(), cyc, 64 * sizeof(long long), hipMemcpyDeviceToHost);
    printf("--- %d threads ---\n", threads);
    const char* names[4] = {"panel_factor<1,no dinv/pmin>", "panel_factor<1>", "variant trailing", "variant chain only"};
    for (int m = 0; m < 4; ++m) printf("%-24s %lld %lld %lld clk per 16-column panel (64 rows)\n", names[m], h[3 * m], h[3 * m + 1], h[3 * m + 2]);
    printf("dependent v_fma_f64          %.1f clk\n", h[12] / 256.0);
    printf("dependent v_mul_f64          %.1f clk\n", h[13] / 256.0);
    printf("dependent v_rsq_f64 + add    %.1f clk\n", h[14] / 64.0);
    printf("readlane pair -> mul -> fma  %.1f clk\n", h[15] / 64.0);
    printf("dependent rsqrt_nr + add     %.1f clk\n", h[16] / 64.0);
    printf("readlane pair -> fma -> fma  %.1f clk\n", h[17] / 64.0);
  }
  return 0;
}
