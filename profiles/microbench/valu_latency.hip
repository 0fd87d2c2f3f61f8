// Micro-benchmark: issue/latency cost (in shader clocks) of the FP64 instruction patterns that bound the
// sequential solver kernels (one wave per SIMD, nothing to hide latency behind).
//   hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency && ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 256
// keep the timed chains between the two counter reads
#define PIN(v) asm volatile("" : "+v"(v) : : "memory")
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ void bench(double* out, long long* cyc, double seed) {
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0000001, z = 0.5;
  long long t0, t1;
  int k = 0;
  double v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  // (a) dependent FMA chain
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_fma(x, y, z);
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  // (b) 8 independent FMA chains
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x + j;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_fma(v[j], y, z);
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
#pragma unroll
  for (int j = 0; j < 8; ++j) x += v[j];
  // (c) dependent rsq chain
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  // (d) readlane (f64 = 2 x b32) feeding an FMA, dependent through x
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) { const double s = readlane_f64(x, i & 63); x = __builtin_fma(x, 0.999, s * 1e-9); }
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  // (e) readlane of an old value + independent FMAs (throughput of the pair)
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x + j;
  const double src = x;
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const double s = readlane_f64(src, (i * 8 + j) & 63); v[j] = __builtin_fma(v[j], y, s); }
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
#pragma unroll
  for (int j = 0; j < 8; ++j) x += v[j];
  // (f) LDS write -> barrier-free read-back round trip (same wave), dependent
  __shared__ double buf[64];
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { buf[lane] = x; x = buf[(lane + 1) & 63] * 0.5 + x; }
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  // (g) s_barrier with 256 threads: cost per barrier
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { __syncthreads(); }
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  // (h) dependent mul chain
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = x * y;
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  // (j) dependent MFMA f64 16x16x4 chain, (k) 4 independent accumulators, (l) LDS->MFMA->LDS round trip
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  {
    f64x4 acc = {x, x, x, x};
    PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
    x = acc[0] + acc[1] + acc[2] + acc[3];
    PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
    if (lane == 0) cyc[9] = t1 - t0;
    f64x4 a4[4] = {{x, x, x, x}, {y, y, y, y}, {z, z, z, z}, {x, y, z, x}};
    PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a4[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, a4[j], 0, 0, 0);
    x = a4[0][0] + a4[1][1] + a4[2][2] + a4[3][3];
    PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
    if (lane == 0) cyc[10] = t1 - t0;
    PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      buf[lane] = x;
      const double av = buf[(lane + 1) & 63], bv = buf[(lane + 17) & 63];
      f64x4 c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, c, 0, 0, 0);
      x = c[0] * 1e-3 + c[3] * 1e-3 + 1.0;
    }
    PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
    if (lane == 0) cyc[11] = t1 - t0;
  }
  // (i) memtime overhead
  PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
  PIN(x); PIN(v[0]); PIN(v[1]); PIN(v[2]); PIN(v[3]); PIN(v[4]); PIN(v[5]); PIN(v[6]); PIN(v[7]); t1 = __builtin_readcyclecounter(); PIN(x);
  if (lane == 0) cyc[k] = t1 - t0; ++k;
  out[threadIdx.x] = x;
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 256 * sizeof(double)); (void)hipMalloc(&cyc, 16 * sizeof(long long));
  for (int threads : {64, 256}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench, dim3(1), dim3(threads), 0, 0, out, cyc, 1.25);
    (void)hipDeviceSynchronize();
    long long h[16]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const long long ov = h[8];
    printf("threads=%d (s_memtime pair overhead %lld)\n", threads, ov);
    printf("  dependent v_fma_f64            : %.1f clk/op\n", double(h[0] - ov) / REP);
    printf("  8 independent v_fma_f64 chains : %.1f clk/op\n", double(h[1] - ov) / REP);
    printf("  dependent v_rsq_f64 + add      : %.1f clk/pair\n", double(h[2] - ov) / REP);
    printf("  dependent readlane_f64 + fma   : %.1f clk/step\n", double(h[3] - ov) / REP);
    printf("  independent readlane_f64 + fma : %.1f clk/step\n", double(h[4] - ov) / REP);
    printf("  LDS write->read round trip     : %.1f clk/step\n", double(h[5] - ov) / 64);
    printf("  __syncthreads                  : %.1f clk\n", double(h[6] - ov) / 64);
    printf("  dependent v_mul_f64            : %.1f clk/op\n", double(h[7] - ov) / REP);
    printf("  dependent mfma_f64_16x16x4     : %.1f clk/op\n", double(h[9] - ov) / 64);
    printf("  4 independent mfma_f64 chains  : %.1f clk/op\n", double(h[10] - ov) / 64);
    printf("  LDS write->read->2 mfma->valu  : %.1f clk/step\n", double(h[11] - ov) / 16);
  }
  return 0;
}
