// Micro-benchmark + check: the 32x32 block factorisation [D; I] -> [L; L⁻ᵀ] the tree levels, the dense reduced solve and
// the blocked reduced factorisation are built on, with Z = L⁻¹X of 80 coupled columns behind it (a tree level's step).
// (a) rounds 1-3: two in-wave 16-column panels + one MFMA tile update + Z = MᵀX as MFMA tiles (four phases, four
// barriers); (b) round 4: block_elim.hpp -- a chief wave on the spine, follower waves with the identity rows and the
// rows of Xᵀ, four columns per step, Z out of the factorisation itself. Prints shader clocks per block and the errors
// of both against a host Cholesky in long double.
//   hipcc --offload-arch=gfx950 -O3 -I ../../calico_amd/csrc block_factor.hip -o /tmp/block_factor && /tmp/block_factor
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "solve_dev.hpp"
#include "block_elim.hpp"
#include "block_split.hpp"

using namespace cal;
#define PIN(v) asm volatile("" : "+v"(v) : : "memory")
constexpr int DLD = 33;

constexpr int XLD = 81;
template <bool NEG, int NK>
__device__ __forceinline__ f64x4 atb_tile_n(const double* P, int ldp, int pc0, const double* Q, int ldq, int qc0, f64x4 acc, int lane) {
  const int l16 = lane & 15, lk = lane >> 4;
  const double* pp = P + lk * ldp + pc0 + l16;
  const double* qq = Q + lk * ldq + qc0 + l16;
  double av[NK], bv[NK];
#pragma unroll
  for (int u = 0; u < NK; ++u) { av[u] = pp[4 * u * ldp]; bv[u] = qq[4 * u * ldq]; }
#pragma unroll
  for (int u = 0; u < NK; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -av[u] : av[u], bv[u], acc, 0, 0, 0);
  return acc;
}
// out_*: [64][32] (L, then L⁻ᵀ) followed by Z [32][80]
template <int MODE, int NOISE = 0>
__global__ __launch_bounds__(512) void bench(const double* D, const double* X, double* out_old, double* out_new, double* out_split, long long* cyc, const double* noise = nullptr) {
  __shared__ double A[64 * DLD], Xs[32 * XLD], Zb[32 * XLD], bcast[128], dinv[80], dumpb[512], chbuf[kElimBufDoubles], cbuf[kSplitDoubles];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lk = lane >> 4;
  double* dump = dumpb + tid;
  double pmin = 1.0, x = 1.0 + lane * 1e-3;
  long long t0, t1;
  const ElimChannel ch = elim_channel(chbuf);
  auto init = [&]() {
    for (int e = tid; e < 1024; e += blockDim.x) {
      const int r = e >> 5, c = e & 31;
      A[r * DLD + c] = D[e];
      A[(32 + r) * DLD + c] = r == c ? 1.0 : 0.0;
    }
    for (int e = tid; e < 32 * 80; e += blockDim.x) { Xs[(e / 80) * XLD + e % 80] = X[e]; Zb[(e / 80) * XLD + e % 80] = -7.0; }
    __syncthreads();
  };
  int k = 4 * MODE, base = 0;
  for (int mode = MODE; mode < MODE + 1; ++mode) {
    for (int rep = 0; rep < 4; ++rep) {
      init();
      if (MODE == 2) { split_reset(cbuf, tid, blockDim.x); __syncthreads(); }
      if (MODE == 1) { elim_reset(ch, tid, blockDim.x); __syncthreads(); }
      PIN(x); t0 = __builtin_readcyclecounter(); PIN(x);
      if (MODE == 0) {
        if (wave == 0) panel_factor<1, false, false>(A, DLD, dinv, bcast, 0, 63, 16, lane, &pmin);
        lds_barrier();
        if (wave < 3) update_tile(A, DLD, 63, 1 + wave, 1, 0, 1, lane, dump);
        lds_barrier();
        if (wave == 0) panel_factor<1, false, false>(A, DLD, dinv, bcast, 16, 63, 16, lane, &pmin);
        lds_barrier();
        const double* M = A + 32 * DLD;
        auto zjob = [&](int jt, int it) {
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          acc = it == 0 ? atb_tile_n<false, 4>(M, DLD, 0, Xs, XLD, 16 * jt, acc, lane) : atb_tile_n<false, 8>(M, DLD, 16, Xs, XLD, 16 * jt, acc, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) Zb[(16 * it + lk + 4 * r) * XLD + 16 * jt + l16] = acc[r];
        };
        if (blockDim.x == 512) { zjob(wave >> 1, wave & 1); if (wave == 0 || wave == 2) zjob(4, wave == 0 ? 1 : 0); }
        lds_barrier();
      } else if (MODE == 1) {
        if (NOISE && wave >= 4) {      // what a tree level's loader waves do beside the elimination: scattered 8-byte loads, then LDS writes
          double v[14];
          const size_t base = size_t(rep) * 65536 + size_t(tid - 256) * 7;
          // NOISE 3: the loads miss every cache (a 1 GiB buffer, one line per load, never the same line twice)
          const size_t far = (size_t(rep) * 0x2000000 + size_t(tid - 256) * 0x20000 + size_t(blockIdx.x)) ;
#pragma unroll
          for (int u = 0; u < 14; ++u) v[u] = NOISE == 3 ? noise[(far + size_t(u) * 0x2000 + 0x40000) & 0x7ffffff] : noise[(base + size_t(u) * 4099) & 0x3ffff];
          double sacc = 0.0;
#pragma unroll
          for (int u = 0; u < 14; ++u) { sacc += v[u]; dumpb[tid] = sacc; }
          if (NOISE >= 2) {           // ... and their arithmetic: integer index work and a few FP64 divisions per entry
            int k2 = tid;
            double q2 = 1.0 + sacc;
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
#pragma unroll
              for (int j = 0; j < 24; ++j) k2 = (k2 * 33 + j) / 6 + (k2 % 7);
              q2 = fmin(fmax(q2 * 1.25, 1e-6), 1e32) / (3.0 + q2 * q2);
            }
            sacc += q2 + k2;
          }
          x += sacc * 1e-300;
        }
        if (wave == 0) {
          long long ts[9];
          elim_chief<1, true>(A, DLD, ch, lane, ts);
          if (lane == 0 && rep == 3) { for (int i = 0; i < 9; ++i) cyc[16 + i] = ts[i] - t0; }
        }
        else if (wave == 1) {
          const ElimTile t[3] = {{Zb, 0, 0, A + 32 * DLD, DLD, 1, 1, nullptr}, {Zb, 0, 0, A + 48 * DLD, DLD, 1, 2, nullptr},
                                 {Xs + 64, 1, XLD, Zb + 64, 1, XLD, 0, nullptr}};
          elim_follow<3>(t, ch, lane);
        } else if (wave == 2 || wave == 3) {
          const int c0 = 32 * (wave - 2);
          const ElimTile t[2] = {{Xs + c0, 1, XLD, Zb + c0, 1, XLD, 0, nullptr}, {Xs + c0 + 16, 1, XLD, Zb + c0 + 16, 1, XLD, 0, nullptr}};
          elim_follow<2>(t, ch, lane);
        }
        base += 8;
        lds_barrier();
      } else if (MODE == 2) {
        long long ts[2] = {0, 0};
        split_factor(A, DLD, cbuf, wave, lane, ts);
        if (lane == 0 && wave < 4 && rep == 3) { cyc[40 + 2 * wave] = ts[0] - t0; cyc[41 + 2 * wave] = ts[1] - t0; }
        lds_barrier();
        if (tid == 0) cyc[32 + rep] = __builtin_readcyclecounter() - t0;
        // Z = MᵀX with M read out of the column buffer: M(k, c) = cbuf[c * kSplitLD + 32 + k]
        auto zjob = [&](int jt, int it) {
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          const int nk = it == 0 ? 4 : 8;
          const double* pp = cbuf + (16 * it + l16) * kSplitLD + 32 + lk;
          const double* qq = Xs + lk * XLD + 16 * jt + l16;
          double av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { av[u] = u < nk ? pp[4 * u] : 0.0; bv[u] = u < nk ? qq[4 * u * XLD] : 0.0; }
#pragma unroll
          for (int u = 0; u < 8; ++u) if (u < nk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) Zb[(16 * it + lk + 4 * r) * XLD + 16 * jt + l16] = acc[r];
        };
        if (blockDim.x == 512) { zjob(wave >> 1, wave & 1); if (wave == 0 || wave == 2) zjob(4, wave == 0 ? 1 : 0); }
        lds_barrier();
      }
      PIN(x); t1 = __builtin_readcyclecounter(); PIN(x);
      if (tid == 0) cyc[k] = t1 - t0;
      ++k;
      __syncthreads();
      double* o = MODE == 0 ? out_old : (MODE == 1 ? out_new : out_split);
      if (MODE == 2) { for (int e = tid; e < 64 * 32; e += blockDim.x) o[e] = cbuf[(e & 31) * kSplitLD + (e >> 5)]; }
      else for (int e = tid; e < 64 * 32; e += blockDim.x) o[e] = A[(e >> 5) * DLD + (e & 31)];
      for (int e = tid; e < 32 * 80; e += blockDim.x) o[2048 + e] = Zb[(e / 80) * XLD + e % 80];
      __syncthreads();
    }
  }
  if (tid == 0) out_old[2048 + 2560] = x + pmin;
}

int main() {
  const int n = 32;
  std::vector<double> D(n * n), L(n * n, 0.0), M(n * n, 0.0);
  // SPD test block shaped like a damped band block: B Bᵀ + diagonal
  std::vector<double> B(n * n);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return double(s >> 11) / double(1ull << 53) - 0.5; };
  for (auto& v : B) v = rnd();
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double a = 0.0;
      for (int k = 0; k < n; ++k) a += B[i * n + k] * B[j * n + k];
      D[i * n + j] = a + (i == j ? 0.05 : 0.0);
    }
  // host Cholesky and inverse
  std::vector<long double> Lh(n * n, 0.0L), Li(n * n, 0.0L);
  for (int j = 0; j < n; ++j) {
    long double d = D[j * n + j];
    for (int k = 0; k < j; ++k) d -= Lh[j * n + k] * Lh[j * n + k];
    Lh[j * n + j] = sqrtl(d);
    for (int i = j + 1; i < n; ++i) {
      long double v = D[i * n + j];
      for (int k = 0; k < j; ++k) v -= Lh[i * n + k] * Lh[j * n + k];
      Lh[i * n + j] = v / Lh[j * n + j];
    }
  }
  for (int c = 0; c < n; ++c)
    for (int i = 0; i < n; ++i) {
      long double v = i == c ? 1.0L : 0.0L;
      for (int k = 0; k < i; ++k) v -= Lh[i * n + k] * Li[k * n + c];
      Li[i * n + c] = v / Lh[i * n + i];
    }
  std::vector<double> X(32 * 80);
  for (auto& v : X) v = rnd();
  std::vector<long double> Zh(32 * 80);
  for (int c = 0; c < 80; ++c)
    for (int i = 0; i < n; ++i) {
      long double v = X[i * 80 + c];
      for (int k = 0; k < i; ++k) v -= Lh[i * n + k] * Zh[k * 80 + c];
      Zh[i * 80 + c] = v / Lh[i * n + i];
    }
  const int NO = 2048 + 2560 + 8;
  double *dD, *dX, *oo, *on, *os; long long* cyc;
  hipMalloc(&dD, n * n * sizeof(double)); hipMalloc(&dX, 32 * 80 * sizeof(double)); hipMalloc(&oo, NO * sizeof(double)); hipMalloc(&on, NO * sizeof(double)); hipMalloc(&os, NO * sizeof(double));
  hipMalloc(&cyc, 64 * sizeof(long long));
  long long* cyc4; hipMalloc(&cyc4, 64 * sizeof(long long)); hipMemset(cyc4, 0, 64 * sizeof(long long));
  long long* cyc3; hipMalloc(&cyc3, 64 * sizeof(long long)); hipMemset(cyc3, 0, 64 * sizeof(long long));
  long long* cyc2; hipMalloc(&cyc2, 64 * sizeof(long long)); hipMemset(cyc2, 0, 64 * sizeof(long long));
  double* dN; hipMalloc(&dN, size_t(0x8000000) * sizeof(double)); hipMemset(dN, 0, size_t(0x8000000) * sizeof(double));
  hipMemcpy(dD, D.data(), n * n * sizeof(double), hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), 32 * 80 * sizeof(double), hipMemcpyHostToDevice);
  for (int threads : {512}) {
    hipMemset(cyc, 0, 64 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(bench<0>, dim3(1), dim3(threads), 0, 0, dD, dX, oo, on, os, cyc);
      hipLaunchKernelGGL(bench<1>, dim3(1), dim3(threads), 0, 0, dD, dX, oo, on, os, cyc);
      hipLaunchKernelGGL((bench<1, 1>), dim3(1), dim3(threads), 0, 0, dD, dX, oo, on, os, cyc2, dN);
      hipLaunchKernelGGL((bench<1, 2>), dim3(1), dim3(threads), 0, 0, dD, dX, oo, on, os, cyc3, dN);
      hipLaunchKernelGGL((bench<1, 3>), dim3(1), dim3(threads), 0, 0, dD, dX, oo, on, os, cyc4, dN);
      hipLaunchKernelGGL(bench<2>, dim3(1), dim3(threads), 0, 0, dD, dX, oo, on, os, cyc);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<long long> h(64);
    std::vector<double> ho(NO), hn(NO), hs(NO);
    hipMemcpy(h.data(), cyc, 64 * sizeof(long long), hipMemcpyDeviceToHost);
    hipMemcpy(ho.data(), oo, NO * sizeof(double), hipMemcpyDeviceToHost);
    hipMemcpy(hn.data(), on, NO * sizeof(double), hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), os, NO * sizeof(double), hipMemcpyDeviceToHost);
    auto err = [&](const std::vector<double>& o, double& eL, double& eM, double& lowM) {
      eL = eM = lowM = 0.0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          if (j <= i) eL = fmax(eL, fabs(o[i * 32 + j] - double(Lh[i * n + j])) / (1.0 + fabs(double(Lh[i * n + j]))));
          // M = L⁻ᵀ: M[i][j] = Li[j][i]
          const double ref = j >= i ? double(Li[j * n + i]) : 0.0;
          if (j >= i) eM = fmax(eM, fabs(o[(32 + i) * 32 + j] - ref) / (1.0 + fabs(ref)));
          else lowM = fmax(lowM, fabs(o[(32 + i) * 32 + j]));
        }
    };
    auto errz = [&](const std::vector<double>& o) {
      double e = 0.0;
      for (int i = 0; i < 32 * 80; ++i) e = fmax(e, fabs(o[2048 + i] - double(Zh[i])) / (1.0 + fabs(double(Zh[i]))));
      return e;
    };
    printf("column split, waves 0..3 [enter own panel, leave]: ");
    for (int i = 0; i < 8; ++i) printf("%lld ", h[40 + i]);
    printf("\n");
    { std::vector<long long> h2(64); hipMemcpy(h2.data(), cyc2, 64 * sizeof(long long), hipMemcpyDeviceToHost);
      printf("chief + followers with four loader waves beside them: %lld %lld %lld %lld clk; chief, from the start: ", h2[4], h2[5], h2[6], h2[7]);
      for (int i = 0; i < 9; ++i) printf("%lld ", h2[16 + i]);
      printf("\n"); }
    { std::vector<long long> h2(64); hipMemcpy(h2.data(), cyc3, 64 * sizeof(long long), hipMemcpyDeviceToHost);
      printf("... loader waves with their arithmetic, too:                %lld %lld %lld %lld clk; chief, from the start: ", h2[4], h2[5], h2[6], h2[7]);
      for (int i = 0; i < 9; ++i) printf("%lld ", h2[16 + i]);
      printf("\n"); }
    { std::vector<long long> h2(64); hipMemcpy(h2.data(), cyc4, 64 * sizeof(long long), hipMemcpyDeviceToHost);
      printf("... loader waves whose loads miss every cache:               %lld %lld %lld %lld clk; chief, from the start: ", h2[4], h2[5], h2[6], h2[7]);
      for (int i = 0; i < 9; ++i) printf("%lld ", h2[16 + i]);
      printf("\n"); }
    printf("chief, clocks from the start: ");
    for (int i = 0; i < 9; ++i) printf("%lld ", h[16 + i]);
    printf("\n");
    double eL, eM, lowM;
    printf("--- %d threads ---\n", threads);
    err(ho, eL, eM, lowM);
    printf("panels + tile + Z (rounds 1-3): %lld %lld %lld %lld clk   err L %.2e  L^-T %.2e  below-diagonal of L^-T %.2e  Z %.2e\n", h[0], h[1], h[2], h[3], eL, eM, lowM, errz(ho));
    err(hn, eL, eM, lowM);
    printf("chief + followers (round 4):    %lld %lld %lld %lld clk   err L %.2e  L^-T %.2e  below-diagonal of L^-T %.2e  Z %.2e\n", h[4], h[5], h[6], h[7], eL, eM, lowM, errz(hn));
    err(hs, eL, eM, lowM);
    printf("column split, 4 waves + Z:      %lld %lld %lld %lld clk (factorisation alone %lld %lld %lld %lld)  err L %.2e  L^-T %.2e  below-diagonal of L^-T %.2e  Z %.2e\n",
           h[8], h[9], h[10], h[11], h[32], h[33], h[34], h[35], eL, eM, lowM, errz(hs));
  }
  return 0;
}
