// Micro-benchmark: what a kernel boundary does to data a consumer reads at its head. One wave walks a chain of DEPENDENT
// loads (a pointer chase through a buffer, one 128-byte line per hop) and reports clocks per hop:
//   cold      : a buffer nothing has touched since it was written by the host copy
//   same-L2   : the buffer was walked by the SAME workgroup index (same XCD) in the previous kernel of the stream
//   other-L2  : the buffer was walked by a workgroup on ANOTHER XCD in the previous kernel
//   in-kernel : second walk inside one kernel (L2 / L1 hit)
//   written   : the previous kernel WROTE the buffer (plain stores) on the same / on another XCD
// Each walk is a kernel of its own with 8 workgroups (one per XCD), only the chosen one works; 100 MHz wall clock and
// shader cycle counter both reported.
//   hipcc --offload-arch=gfx950 -O3 first_touch.hip -o bin/first_touch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

constexpr int kHops = 64;
constexpr int kLineInts = 32;     // 128-byte lines

__global__ void walk_kernel(const int* buf, int wg_active, int twice, long long* out) {
  if (int(blockIdx.x) != wg_active || threadIdx.x != 0) return;
  int idx = 0;
  const long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int h = 0; h < kHops; ++h) idx = buf[size_t(idx) * kLineInts];
  const long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  long long c2 = c1; unsigned long long t2 = t1;
  int idx2 = 0;
  if (twice) {
    for (int h = 0; h < kHops; ++h) idx2 = buf[size_t(idx2) * kLineInts];
    c2 = __builtin_readcyclecounter(); t2 = __builtin_amdgcn_s_memrealtime();
  }
  out[0] = c1 - c0; out[1] = (long long)(t1 - t0); out[2] = c2 - c1; out[3] = (long long)(t2 - t1); out[4] = idx + idx2;
}
__global__ void write_kernel(int* buf, const int* next, int n_lines, int wg_active) {
  if (int(blockIdx.x) != wg_active) return;
  for (int i = threadIdx.x; i < n_lines; i += blockDim.x) buf[size_t(i) * kLineInts] = next[i];
}
__global__ void filler_kernel(int* p) { if (threadIdx.x == 1000) p[0] = 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n_lines = 4096;          // 512 KB per buffer: the chain visits 64 of them
  std::vector<int> perm(n_lines), next(n_lines), host(size_t(n_lines) * kLineInts, 0);
  std::iota(perm.begin(), perm.end(), 0);
  std::mt19937 rng(7); std::shuffle(perm.begin() + 1, perm.end(), rng);
  for (int i = 0; i < n_lines; ++i) next[perm[i]] = perm[(i + 1) % n_lines];
  for (int i = 0; i < n_lines; ++i) host[size_t(i) * kLineInts] = next[i];
  const int NB = 24;
  std::vector<int*> bufs(NB);
  for (int b = 0; b < NB; ++b) { CK(hipMalloc(&bufs[b], host.size() * 4)); CK(hipMemcpy(bufs[b], host.data(), host.size() * 4, hipMemcpyHostToDevice)); }
  int* d_next; CK(hipMalloc(&d_next, n_lines * 4)); CK(hipMemcpy(d_next, next.data(), n_lines * 4, hipMemcpyHostToDevice));
  long long* d_out; CK(hipMalloc(&d_out, 64)); long long h[5];
  int* d_dummy; CK(hipMalloc(&d_dummy, 64));
  auto report = [&](const char* name) {
    (void)hipStreamSynchronize(s); (void)hipMemcpy(h, d_out, 40, hipMemcpyDeviceToHost);
    std::printf("%-52s %7.0f clocks / hop  %6.3f us / hop", name, double(h[0]) / kHops, double(h[1]) / 100.0 / kHops);
    if (h[2]) std::printf("   | second walk in the same kernel: %6.0f clocks, %6.3f us / hop", double(h[2]) / kHops, double(h[3]) / 100.0 / kHops);
    std::printf("\n");
  };
  int b = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(filler_kernel, dim3(8), dim3(64), 0, s, d_dummy);
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 1, d_out); report("cold (host copy only), then again in the kernel"); ++b;
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out);
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out); report("walked by the same XCD in the previous kernel"); ++b;
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 3, 0, d_out);
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out); report("walked by another XCD in the previous kernel"); ++b;
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out);
    hipLaunchKernelGGL(filler_kernel, dim3(8), dim3(64), 0, s, d_dummy);
    hipLaunchKernelGGL(filler_kernel, dim3(8), dim3(64), 0, s, d_dummy);
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out); report("walked by the same XCD three kernels ago"); ++b;
    hipLaunchKernelGGL(write_kernel, dim3(8), dim3(256), 0, s, bufs[b], d_next, n_lines, 0);
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out); report("WRITTEN by the same XCD in the previous kernel"); ++b;
    hipLaunchKernelGGL(write_kernel, dim3(8), dim3(256), 0, s, bufs[b], d_next, n_lines, 3);
    hipLaunchKernelGGL(walk_kernel, dim3(8), dim3(64), 0, s, bufs[b], 0, 0, d_out); report("WRITTEN by another XCD in the previous kernel"); ++b;
  }
  return 0;
}
