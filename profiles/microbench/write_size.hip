// write_size.hip -- calibration of rocprofv3's WRITE_SIZE (and FETCH_SIZE) on gfx950 against KNOWN byte counts, in the
// access patterns of the evaluation chain ("16.5 MB reported for 6 MB of triangles -- calibrate").
//   profiles/microbench/bin/write_size            (run under: rocprofv3 --kernel-trace --pmc WRITE_SIZE ...)
// Kernels (every one writes or reads exactly `bytes` useful bytes, printed by the host):
//   w_plain_full      contiguous 8-byte stores, whole 128-byte lines (coalesced)
//   w_sc1_full        the same with write-through stores (__hip_atomic_store, agent scope: what block_store uses)
//   w_plain_tri       upper triangles of n1 x n1 blocks stored row-major in a FULL square (row i: entries i..n1-1), n1 = 51:
//                     the cell blocks' layout -- rows start at arbitrary 8-byte offsets, lines are written partially
//   w_sc1_tri         the same with write-through stores
//   w_sc1_tri_tiles   the same entries, in the ORDER the cell kernel's MFMA tiles write them (a lane holds four rows of one
//                     column of a 16x16 tile: sixteen 8-byte stores of a wave land in sixteen different rows)
//   r_tri             reads the triangles back (gather's value loads), 8 bytes per lane, row-major
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void w_plain_full(double* p, size_t n) { const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; if (i < n) p[i] = double(i); }
__global__ void w_sc1_full(double* p, size_t n) { const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; if (i < n) st_sc1(p + i, double(i)); }
// one workgroup (256 threads) per block of n1 x n1; entry e of the packed triangle enumeration -> (i, j)
template <bool SC1>
__global__ void w_tri(double* p, int n1) {
  double* out = p + size_t(blockIdx.x) * n1 * n1;
  const int n_tri = n1 * (n1 + 1) / 2;
  for (int e = threadIdx.x; e < n_tri; e += blockDim.x) {
    int i = 0, rem = e;
    while (rem >= n1 - i) { rem -= n1 - i; ++i; }
    const int j = i + rem;
    if (SC1) st_sc1(out + size_t(i) * n1 + j, double(e)); else out[size_t(i) * n1 + j] = double(e);
  }
}
// the order of stage_b_mfma's stores: tile (I, J) of 16 x 16, lane (lc16, lk) stores rows 16 I + lk + 4 r, column 16 J + lc16
__global__ void w_sc1_tri_tiles(double* p, int n1) {
  double* out = p + size_t(blockIdx.x) * n1 * n1;
  const int NT = (n1 + 15) / 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc16 = lane & 15, lk = lane >> 4;
  int t = 0;
  for (int I = 0; I < NT; ++I)
    for (int J = I; J < NT; ++J, ++t) {
      if ((t & 1) != wave) continue;        // two waves share the tiles (blockDim = 128)
      for (int r = 0; r < 4; ++r) {
        const int gi = 16 * I + lk + 4 * r, gj = 16 * J + lc16;
        if (gi <= gj && gj < n1) st_sc1(out + size_t(gi) * n1 + gj, double(gi));
      }
    }
}
__global__ void r_tri(const double* p, int n1, double* sink) {
  const double* in = p + size_t(blockIdx.x) * n1 * n1;
  const int n_tri = n1 * (n1 + 1) / 2;
  double s = 0.0;
  for (int e = threadIdx.x; e < n_tri; e += blockDim.x) {
    int i = 0, rem = e;
    while (rem >= n1 - i) { rem -= n1 - i; ++i; }
    s += in[size_t(i) * n1 + i + rem];
  }
  if (s == 12345.678) sink[0] = s;
}

int main() {
  const int n1 = 51, n_blocks = 4608;                   // 4608 x 1326 x 8 B = 48.9 MB of triangle entries (squares: 95.9 MB, beyond an L2, inside the Infinity Cache)
  const size_t sq = size_t(n_blocks) * n1 * n1, tri = size_t(n_blocks) * (n1 * (n1 + 1) / 2);
  const size_t n_full = size_t(6) << 20;                // 6 Mi doubles = 50.3 MB
  double *a, *b, *sink;
  hipMalloc(&a, sq * 8); hipMalloc(&b, n_full * 8); hipMalloc(&sink, 64);
  hipMemset(a, 0, sq * 8); hipMemset(b, 0, n_full * 8);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(w_plain_full, dim3(unsigned((n_full + 255) / 256)), dim3(256), 0, 0, b, n_full);
    hipLaunchKernelGGL(w_sc1_full, dim3(unsigned((n_full + 255) / 256)), dim3(256), 0, 0, b, n_full);
    hipLaunchKernelGGL(w_tri<false>, dim3(n_blocks), dim3(256), 0, 0, a, n1);
    hipLaunchKernelGGL(w_tri<true>, dim3(n_blocks), dim3(256), 0, 0, a, n1);
    hipLaunchKernelGGL(w_sc1_tri_tiles, dim3(n_blocks), dim3(128), 0, 0, a, n1);
    hipLaunchKernelGGL(r_tri, dim3(n_blocks), dim3(256), 0, 0, a, n1, sink);
    hipDeviceSynchronize();
  }
  std::printf("useful bytes per launch: w_plain_full / w_sc1_full %zu | w_tri<0> / w_tri<1> / w_sc1_tri_tiles / r_tri %zu (the squares they lie in: %zu)\n",
              n_full * 8, tri * 8, sq * 8);
  return 0;
}
