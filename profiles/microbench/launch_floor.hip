// Micro-benchmark: what one kernel of a dependent chain costs on the device timeline when it does nothing, and what
// that depends on: kernel-argument size, dynamic LDS, workgroup size, grid size. N kernels back to back in one stream,
// wall time / N (the stream is kept full, so this is the device-side floor, not the host launch cost).
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Small { int* p; int n; };
struct Big { int* p; int n; double pad[62]; };   // ~512 bytes

template <class A> __global__ void empty_kernel(A a) {
  extern __shared__ double lds[];
  if (a.n == -1) { lds[threadIdx.x] = 1.0; a.p[0] = int(lds[0]); }   // never true: keeps the arguments and the LDS alive
}
__global__ __launch_bounds__(64) void scratch_kernel(Small a) {      // a private array indexed at run time: needs scratch
  double v[64];
  for (int i = 0; i < 64; ++i) v[i] = double(i + a.n);
  if (a.n == -1) a.p[0] = int(v[a.p[1] & 63]);
}

// the same 512 bytes of arguments, all of them read: passed by value (kernarg segment) or through one pointer to a
// device-resident copy
__global__ void use_args_by_value(Big a) {
  double t = 0.0;
  for (int i = 0; i < 62; ++i) t += a.pad[i];
  if (t == -1.0) a.p[0] = 1;
}
__global__ void use_args_by_pointer(const Big* __restrict__ ap) {
  const Big a = *ap;
  double t = 0.0;
  for (int i = 0; i < 62; ++i) t += a.pad[i];
  if (t == -1.0) a.p[0] = 1;
}

template <class F> double chain(F launch, int n) {
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 50; ++i) launch();
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) launch();
  (void)hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; (void)hipMalloc(&d, 64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel<Small>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel<Big>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  Small sa{d, 0}; Big ba{d, 0, {}};
  const int N = 2000;
  std::printf("64 wg x 64, small args, no LDS      : %.2f us per kernel\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Small>, dim3(64), dim3(64), 0, s, sa); }, N));
  std::printf("64 wg x 64, 512-byte args           : %.2f us\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Big>, dim3(64), dim3(64), 0, s, ba); }, N));
  std::printf("64 wg x 64, 100 KB dynamic LDS      : %.2f us\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Small>, dim3(64), dim3(64), 100 * 1024, s, sa); }, N));
  std::printf("1 wg x 512, 100 KB LDS, big args    : %.2f us\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Big>, dim3(1), dim3(512), 100 * 1024, s, ba); }, N));
  std::printf("2500 wg x 256, small args           : %.2f us\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Small>, dim3(2500), dim3(256), 0, s, sa); }, N));
  std::printf("870 wg x 64, 29 KB LDS, big args    : %.2f us\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Big>, dim3(870), dim3(64), 29 * 1024, s, ba); }, N));
  std::printf("64 wg x 64, scratch (private array) : %.2f us\n", chain([&] { hipLaunchKernelGGL(scratch_kernel, dim3(64), dim3(64), 0, s, sa); }, N));
  std::printf("alternating two different kernels   : %.2f us\n", chain([&] { hipLaunchKernelGGL(empty_kernel<Small>, dim3(64), dim3(64), 0, s, sa); hipLaunchKernelGGL(empty_kernel<Big>, dim3(64), dim3(64), 0, s, ba); }, N) / 2);
  Big* dba; (void)hipMalloc(&dba, sizeof(Big)); (void)hipMemcpy(dba, &ba, sizeof(Big), hipMemcpyHostToDevice);
  std::printf("512-byte args read, by value         : %.2f us\n", chain([&] { hipLaunchKernelGGL(use_args_by_value, dim3(64), dim3(64), 0, s, ba); }, N));
  std::printf("512-byte args read, through a pointer: %.2f us\n", chain([&] { hipLaunchKernelGGL(use_args_by_pointer, dim3(64), dim3(64), 0, s, dba); }, N));
  return 0;
}
