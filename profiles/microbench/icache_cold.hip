// Micro-benchmark: what straight-line code costs the FIRST time a kernel runs through it. One wave (or eight waves of one
// workgroup) executes a block of N independent 8-byte VALU instructions (v_fma_f64 on registers: no memory operands) twice
// inside one launch; the first pass fetches the code (instruction cache cold after the kernel boundary: other kernels
// ran in between), the second pass finds it in the instruction cache. Reports clocks per instruction of both passes,
// for code sizes from 1 KB to 32 KB, with and without another kernel (128 KB of other code) run between the launches.
//   hipcc --offload-arch=gfx950 -O3 icache_cold.hip -o bin/icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP_BODY(n) asm volatile(".rept " #n "\n v_fma_f64 %0, %1, %1, %0\n .endr\n" : "+v"(x) : "v"(y))

template <int N>
__global__ void straight_kernel(double* out, long long* clk, int passes) {
  double x = out[threadIdx.x], y = 1.0000001;
  long long t[4];
#pragma unroll 1
  for (int p = 0; p < passes; ++p) {
    const long long c0 = __builtin_readcyclecounter();
    if (N == 128) REP_BODY(128);
    if (N == 512) REP_BODY(512);
    if (N == 1024) REP_BODY(1024);
    if (N == 2048) REP_BODY(2048);
    if (N == 4096) REP_BODY(4096);
    const long long c1 = __builtin_readcyclecounter();
    if (p < 4) t[p] = c1 - c0;
  }
  out[threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0) { clk[(threadIdx.x >> 6) * 4 + 0] = t[0]; clk[(threadIdx.x >> 6) * 4 + 1] = t[1]; }
}
// other code between the launches: evicts the instruction cache (16384 instructions = 128 KB)
__global__ void other_kernel(double* out) {
  double x = out[threadIdx.x], y = 0.9999999;
  REP_BODY(4096); REP_BODY(4096); REP_BODY(4096); REP_BODY(4096);
  out[threadIdx.x] = x;
}

template <int N>
void run(double* d, long long* dc, int threads, bool evict) {
  long long h[32];
  double best0 = 1e30, best1 = 1e30, sum0 = 0;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    if (evict) hipLaunchKernelGGL(other_kernel, dim3(256), dim3(64), 0, 0, d + 1024);
    hipLaunchKernelGGL(straight_kernel<N>, dim3(1), dim3(threads), 0, 0, d, dc, 2);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    if (r == 0) continue;       // (the very first launch also loads the code object)
    best0 = h[0] < best0 ? h[0] : best0; best1 = h[1] < best1 ? h[1] : best1; sum0 += h[0];
  }
  std::printf("%5d instructions (%5.1f KB), %d wave(s), %s: first pass %7.0f clocks (%5.2f / instruction; mean %7.0f), second pass %7.0f (%5.2f / instruction)\n",
              N, N * 8 / 1024.0, threads / 64, evict ? "other kernel between launches" : "back to back              ", best0, best0 / N, sum0 / (reps - 1), best1, best1 / N);
}
int main() {
  double* d; long long* dc;
  (void)hipMalloc(&d, 1 << 20); (void)hipMemset(d, 0, 1 << 20); (void)hipMalloc(&dc, 1024);
  for (int threads : {64, 512}) for (int ev = 0; ev < 2; ++ev) {
    run<128>(d, dc, threads, ev); run<512>(d, dc, threads, ev); run<1024>(d, dc, threads, ev); run<2048>(d, dc, threads, ev); run<4096>(d, dc, threads, ev);
  }
  return 0;
}
