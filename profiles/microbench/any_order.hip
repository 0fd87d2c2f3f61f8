// Micro-benchmark: does a kernel launched with hipExtAnyOrderLaunch (AQL packet without the barrier bit) start while its
// predecessor in the SAME stream is still running on gfx950 / ROCm 7.2, and are the workgroups of the two dispatches
// started in packet order?  (hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for the module launch API.)
//   test 1: A = 1 workgroup spinning 60 us, B = 1 workgroup that files its start time: B.start - A.start.
//   test 2: A = more workgroups than the chip holds at once (each spins 20 us), B = 1 workgroup: when does B start
//           relative to the start of A's LAST workgroup (in-order dispatch <=> B never starts before it).
//   test 3: a dependent chain of N kernels, each waiting in-kernel on a flag its predecessor stores at its end
//           (sc1 store / sc1 poll), launched with and without the barrier bit: time per kernel.
//   hipcc --offload-arch=gfx950 -O3 any_order.hip -o bin/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void spin_kernel(unsigned long long* start, unsigned long long* end, int spin_100mhz_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) start[blockIdx.x] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_100mhz_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

// one link of a dependent chain: wait for flag >= my - 1 (unless first), do `work` ticks, store flag = my
__global__ void link_kernel(int* flag, int my, int wait_in_kernel, int work_ticks, unsigned long long* t_start, unsigned long long* t_go) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (wait_in_kernel && threadIdx.x == 0) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < my - 1) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t1 < (unsigned long long)work_ticks) __builtin_amdgcn_s_sleep(2);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    t_start[my] = t0; t_go[my] = t1;
    __hip_atomic_store(flag, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int MAXWG = 1 << 16;
  unsigned long long *d_start, *d_end, *d_bs, *d_be;
  CK(hipMalloc(&d_start, MAXWG * 8)); CK(hipMalloc(&d_end, MAXWG * 8)); CK(hipMalloc(&d_bs, 64)); CK(hipMalloc(&d_be, 64));
  std::vector<unsigned long long> hs(MAXWG), he(MAXWG);
  unsigned long long bs = 0, be = 0;
  for (int flags = 0; flags <= 1; ++flags) {
    // test 1
    for (int rep = 0; rep < 3; ++rep) {
      hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d_start, d_end, 6000);
      hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, d_bs, d_be, 100);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(hs.data(), d_start, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bs, d_bs, 8, hipMemcpyDeviceToHost));
      std::printf("flags %d test 1: B starts %.2f us after A starts (A spins 60 us)\n", flags, (double(bs) - double(hs[0])) / 100.0);
    }
    // test 2: 1024-thread workgroups with 64 KB LDS: 2 per CU -> 512 at a time; 2048 of them = 4 rounds of 20 us
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
      const int nA = 2048;
      hipExtLaunchKernelGGL(spin_kernel, dim3(nA), dim3(1024), 64 * 1024, s, nullptr, nullptr, 0, d_start, d_end, 2000);
      hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, d_bs, d_be, 100);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(hs.data(), d_start, nA * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(he.data(), d_end, nA * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&bs, d_bs, 8, hipMemcpyDeviceToHost));
      const unsigned long long first = *std::min_element(hs.begin(), hs.begin() + nA), last_start = *std::max_element(hs.begin(), hs.begin() + nA),
                               last_end = *std::max_element(he.begin(), he.begin() + nA);
      std::printf("flags %d test 2: A's workgroups start over %.1f us, A ends at %.1f us; B starts at %.1f us (%.1f us after A's last start)\n", flags,
                  (double(last_start) - double(first)) / 100.0, (double(last_end) - double(first)) / 100.0, (double(bs) - double(first)) / 100.0,
                  (double(bs) - double(last_start)) / 100.0);
    }
  }
  // test 3: chain of dependent kernels
  int* d_flag; CK(hipMalloc(&d_flag, 64));
  unsigned long long *d_ts, *d_tg; CK(hipMalloc(&d_ts, 8 * 4096)); CK(hipMalloc(&d_tg, 8 * 4096));
  const int N = 600;
  std::vector<unsigned long long> ts(N + 1), tg(N + 1);
  for (int work : {0, 500, 2000}) {
    for (int mode = 0; mode < 3; ++mode) {      // 0: barrier bit (plain stream order); 1: barrier bit + in-kernel wait (cost of the wait alone); 2: no barrier bit + in-kernel wait
      for (int wgs : {1, 256}) {
        CK(hipMemsetAsync(d_flag, 0, 4, s)); CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= N; ++i)
          hipExtLaunchKernelGGL(link_kernel, dim3(wgs), dim3(256), 0, s, nullptr, nullptr, mode == 2 ? 1 : 0, d_flag, i, mode >= 1 ? 1 : 0, work, d_ts, d_tg);
        CK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        CK(hipMemcpy(ts.data(), d_ts, 8 * (N + 1), hipMemcpyDeviceToHost)); CK(hipMemcpy(tg.data(), d_tg, 8 * (N + 1), hipMemcpyDeviceToHost));
        // steady state: go-to-go distance of consecutive links over the second half of the chain
        const double per = (double(tg[N]) - double(tg[N / 2])) / 100.0 / (N - N / 2);
        std::printf("test 3: work %5.1f us, %3d wg, mode %d (%s): %.2f us per link on the device (host wall %.2f us per link)\n", work / 100.0, wgs, mode,
                    mode == 0 ? "barrier bit" : mode == 1 ? "barrier bit + flag wait" : "NO barrier bit + flag wait", per, us / N);
      }
    }
  }
  return 0;
}
