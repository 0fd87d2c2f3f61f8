// Micro-benchmark: what a fork / join between two HIP streams costs on the device timeline, against plain kernel
// boundaries in one stream. Chain A: K(main) K(main) K(main) K(main), each ~10 us. Chain B: K(main) -> [K(main) || K(aux)]
// -> K(main) with event record / stream-wait on both sides. If B is not ~10 us shorter than A the cross-stream
// dependencies cost more than the concurrency buys.
//   hipcc --offload-arch=gfx950 -O3 fork_join.hip -o /tmp/fork_join && /tmp/fork_join
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(long long clocks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < clocks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

int main() {
  hipStream_t m, a;
  hipStreamCreateWithFlags(&m, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipEvent_t e1, e2;
  hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  int* sink; hipMalloc(&sink, 4);
  const long long c10 = 1000;   // wall_clock64 runs at 100 MHz: 1000 ticks = 10 us
  auto run = [&](int mode, int reps) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, m, c10, sink);
      if (mode == 0) {
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, m, c10, sink);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, m, c10, sink);
      } else {
        hipEventRecord(e1, m);
        hipStreamWaitEvent(a, e1, 0);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, a, c10, sink);
        hipEventRecord(e2, a);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, m, c10, sink);
        hipStreamWaitEvent(m, e2, 0);
      }
      hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, m, c10, sink);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  };
  run(0, 20); run(1, 20);
  for (int k = 0; k < 3; ++k)
    std::printf("one stream, 4 kernels of 10 us: %.1f us per round | fork/join (2 of them side by side): %.1f us per round\n", run(0, 200), run(1, 200));
  return 0;
}
