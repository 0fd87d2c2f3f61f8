// mfma_f64_rate.hip -- what one v_mfma_f64_16x16x4_f64 costs on gfx950 when (a) one wave issues them alone on a CU, (b) waves on
// all four SIMDs of a CU do, (c) every CU of the chip does. DESIGN.md quotes "an FP64 MFMA costs ~110-130 clocks in the
// evaluation's loops against 64 alone on a CU": this separates "the other SIMDs of the CU" from "the whole chip under FP64 load".
//   profiles/microbench/bin/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kIter = 2048, kAcc = 8;
__global__ void mfma_rate(double* out, unsigned long long* ticks, int chain) {
  f64x4 acc[kAcc];
  for (int i = 0; i < kAcc; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double a = double(threadIdx.x & 15) * 1e-3, b = double(threadIdx.x >> 4) * 1e-3;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (chain == 4 || chain == 5) {   // stage B's shape: ten accumulators, four operand registers in the pairs of the upper triangle; 5: the four operands loaded from LDS per k-step
    __shared__ double sh[4 * 68];
    for (int i = threadIdx.x; i < 4 * 68; i += blockDim.x) sh[i] = 1e-3 * double(i % 13);
    __syncthreads();
    f64x4 c8 = f64x4{0, 0, 0, 0}, c9 = c8;
    double o0 = a, o1 = b, o2 = a + b, o3 = a - b;
    const double* sp = sh + (threadIdx.x & 3);
    for (int it = 0; it < kIter * kAcc / 10; ++it) {
      if (chain == 5) { o0 = sp[(it & 15) * 4]; o1 = sp[68 + (it & 15) * 4]; o2 = sp[136 + (it & 15) * 4]; o3 = sp[204 + (it & 15) * 4]; }
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o0, o0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o0, o1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(o0, o2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(o0, o3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(o1, o1, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(o1, o2, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(o1, o3, acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(o2, o2, acc[7], 0, 0, 0);
      c8 = __builtin_amdgcn_mfma_f64_16x16x4f64(o2, o3, c8, 0, 0, 0);
      c9 = __builtin_amdgcn_mfma_f64_16x16x4f64(o3, o3, c9, 0, 0, 0);
    }
    acc[0] += c8 + c9;
  } else if (chain == 2) {   // ONE accumulator, and the operand of every product selected by two VALU instructions (v_cndmask) in front of it: the frames' loop
    int rows = int(ticks[0] & 127) + 64;     // (not known at compile time)
    for (int it = 0; it < kIter * kAcc; ++it) {
      const double x = ((threadIdx.x >> 4) + (it & 31) * 4 < rows) ? a : 0.0;
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc[0], 0, 0, 0);
    }
  } else if (chain == 3) {   // the same selections, eight at a time in front of eight products
    int rows = int(ticks[0] & 127) + 64;
    for (int it = 0; it < kIter; ++it) {
      double x[kAcc];
#pragma unroll
      for (int i = 0; i < kAcc; ++i) x[i] = ((threadIdx.x >> 4) + ((it * kAcc + i) & 31) * 4 < rows) ? a : 0.0;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < kAcc; ++i) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[i], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (chain) {        // ONE accumulator: every MFMA depends on the one before
    for (int it = 0; it < kIter * kAcc; ++it) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
  } else {
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
      for (int i = 0; i < kAcc; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int i = 0; i < kAcc; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
  double* out; unsigned long long* ticks;
  hipMalloc(&out, 2048 * 256 * 8); hipMalloc(&ticks, 2048 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n_mfma = kIter * kAcc;
  std::printf("v_mfma_f64_16x16x4_f64: %d per wave. 'counter' = __builtin_readcyclecounter() ticks per MFMA (median wave), 'wall' = ns per MFMA per wave from HIP events; 2048 flop each\n", n_mfma);
  struct Cfg { int wgs, waves, chain; const char* what; };
  const Cfg cfgs[] = {{1, 1, 0, "one wave alone on a CU, 8 accumulators"}, {1, 1, 1, "one wave alone, ONE accumulator (dependent chain)"},
                      {1, 1, 2, "one wave alone, one accumulator, operand selected (2 x v_cndmask) in front of EVERY product"},
                      {1, 1, 3, "one wave alone, one accumulator, eight operands selected, then eight products"},
                      {1, 1, 4, "one wave alone, stage B's shape: ten accumulators, operands in registers (A != B)"},
                      {1, 1, 5, "one wave alone, stage B's shape, the four operands loaded from LDS per k-step"},
                      {1, 4, 0, "four waves (one per SIMD) of one CU"}, {1, 8, 0, "eight waves (two per SIMD) of one CU"},
                      {256, 1, 0, "one wave on each of 256 CUs"}, {256, 4, 0, "four waves on each of 256 CUs (the chip's FP64 matrix peak)"},
                      {1024, 4, 0, "1024 workgroups of four waves"}};
  for (const Cfg& c : cfgs) {
    float best = 1e30f; unsigned long long med = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma_rate, dim3(c.wgs), dim3(64 * c.waves), 0, 0, out, ticks, c.chain);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const int nw = c.wgs * c.waves;
    static unsigned long long h[8192];
    hipMemcpy(h, ticks, size_t(nw) * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < nw; ++i) for (int j = i + 1; j < nw && i <= nw / 2; ++j) if (h[j] < h[i]) { unsigned long long t = h[i]; h[i] = h[j]; h[j] = t; }
    med = h[nw / 2];
    const double waves_in_flight_rounds = c.wgs > 256 ? double(c.wgs) / 256.0 : 1.0;
    std::printf("%-62s counter %7.1f  wall %7.1f ns  -> %6.2f TFLOP/s on the device\n", c.what, double(med) / n_mfma,
                double(best) * 1e6 / n_mfma / waves_in_flight_rounds, double(nw) * n_mfma * 2048.0 / (double(best) * 1e-3) / 1e12);
  }
  return 0;
}
