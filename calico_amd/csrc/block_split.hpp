// block_split.hpp — the 32x32 block factorisation D = L Lᵀ with M = L⁻ᵀ riding along, split by COLUMNS over four waves
// on the four SIMDs of a CU (round 4).
//
// Rounds 1-3 factored the augmented block [D; I] (64 rows, one lane per row) as two in-wave 16-column panels with an
// MFMA tile update between them: panel | barrier | tile | barrier | panel. One wave issues every instruction of a
// panel -- 31 per column, of which the dependency chain (pivot -> v_rsq_f64 + Newton -> broadcast -> scale -> next
// pivot, ~70 clocks) needs about ten; the rest are the trailing updates of the panel's other columns, and they sit in
// the same instruction stream. ~3.0k clocks per panel alone on a SIMD, 3.9-5.9k inside the kernels.
//
// Here wave w (w = 0..3, one per SIMD) owns columns [8w, 8w + 8) of all 64 rows (lane = row, eight registers):
//   * it first CATCHES UP: for every earlier column j, published by its owner in LDS, a(row, c) -= L(row, j) L(8w + c, j)
//     (one 64-lane read of the column, the eight multipliers as wave-uniform reads, eight FMAs), running behind the
//     producers while they work;
//   * then it runs the chain for its own eight columns -- the same chain as panel_factor (every lane follows its own
//     would-be pivot, the pivot lane's 1/sqrt goes round by one v_readlane pair), but with at most seven trailing
//     columns instead of fifteen -- and PUBLISHES each column as it is scaled.
// No barrier, no flag: the column buffer is filled with a sentinel (a NaN no arithmetic produces) before the
// factorisation and a consumer polls its own lane's entry of the column; LDS executes instructions in order, so once
// all 64 entries of a column have been seen, the multipliers read behind them are there, too. The chain moves from
// wave to wave three times (each hand-over costs an LDS write -> read round trip and eight FMAs).
// The factor never goes back to the block's own storage: the column buffer IS the result,
//     cb[j * kSplitLD + r] = L(r, j)  (r >= j; above the diagonal: undefined but finite),   cb[j * kSplitLD + 32 + i] = M(i, j) = L⁻ᵀ(i, j),
// which is what the consumers (Z = MᵀX tiles, the filing of L⁻ᵀ, the dense solve's L) read, transposed.
// A bad pivot is not patched: NaN propagates (it is not the sentinel) and is caught by the update stage.
#pragma once
#include "solve_dev.hpp"

namespace cal {

constexpr int kSplitLD = 65;                         // column stride of the column buffer (64 rows + 1: conflict-free both ways)
constexpr int kSplitDoubles = 32 * kSplitLD;
constexpr unsigned long long kSplitSentinel = 0x7FF8C0DEC0DEC0DEull;

// all threads of the workgroup (or any subset that covers the buffer); the caller's barrier follows
DEVI void split_reset(double* cb, int tid, int nthreads) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(cb);
  for (int e = tid; e < kSplitDoubles; e += nthreads) p[e] = kSplitSentinel;
}

// Wave W of the four. D: the block, rows 0..31, row stride LD (LDS); both triangles are read (the upper one only feeds
// entries nobody uses, but it must be finite).
template <int W>
DEVI void split_factor_wave(const double* D, int LD, double* cb, int lane, long long* ts = nullptr) {
  constexpr int c0 = 8 * W;
  double av[8];
  {
    double t[8];
    const double* src = D + (lane & 31) * LD + c0;
#pragma unroll
    for (int c = 0; c < 8; ++c) t[c] = src[c];
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(t[c]));      // (unconditional loads: a select, not a branch)
#pragma unroll
    for (int c = 0; c < 8; ++c) av[c] = lane < 32 ? t[c] : (lane - 32 == c0 + c ? 1.0 : 0.0);
  }
  unsigned long long* const cbu = reinterpret_cast<unsigned long long*>(cb);
  // ---- catch up with the columns of the waves before this one: column j + 1 is requested before column j is applied ----
  if (W > 0) {
    const unsigned long long* own = cbu + lane;
    const double* mul = cb + c0;
    unsigned long long v = __hip_atomic_load(own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    double m[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = mul[c];
#pragma unroll 1
    for (int j = 0; j < c0; ++j) {
      while (__builtin_amdgcn_ballot_w64(v == kSplitSentinel) != 0) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) m[c] = mul[c];
      }
      const double lj = __longlong_as_double((long long)v);
      double mc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) mc[c] = m[c];
      asm volatile("" ::: "memory");
      own += kSplitLD; mul += kSplitLD;
      if (j + 1 < c0) {
        v = __hip_atomic_load(own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) m[c] = mul[c];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) av[c] = __builtin_fma(-lj, mc[c], av[c]);
    }
  }
  // ---- this wave's eight columns ----
  if (ts) ts[0] = __builtin_readcyclecounter();
  double lprev = 0.0;
  double pown = av[0];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    // multipliers of column jj-1 for the columns from jj+2 on: requested before the pivot chain starts, consumed after it
    double lb[8];
    if (jj > 0) {
#pragma unroll
      for (int c = jj + 2; c < 8; ++c) lb[c] = cb[(c0 + jj - 1) * kSplitLD + c0 + c];
    }
    const double rs_own = rsqrt_nr(pown);
    const double rs = readlane_f64(rs_own, c0 + jj);
    const double l = av[jj] * rs;
    if (jj + 1 < 8) pown = __builtin_fma(-l, l, av[jj + 1]);    // the next pivot, in the lane that owns it
    __hip_atomic_store(cbu + (c0 + jj) * kSplitLD + lane, (unsigned long long)__double_as_longlong(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (jj + 1 < 8) av[jj + 1] = __builtin_fma(-l, readlane_f64(l, c0 + jj + 1), av[jj + 1]);
    if (jj + 2 < 8) av[jj + 2] = __builtin_fma(-l, readlane_f64(l, c0 + jj + 2), av[jj + 2]);
    if (jj > 0) {
#pragma unroll
      for (int c = jj + 2; c < 8; ++c) av[c] = __builtin_fma(-lprev, lb[c], av[c]);
    }
    lprev = l;
  }
  if (ts) ts[1] = __builtin_readcyclecounter();
}

// Call with the four waves that factor (wave index 0..3 of the workgroup, one per SIMD); the others go past.
DEVI void split_factor(const double* D, int LD, double* cb, int wave, int lane, long long* ts = nullptr) {
  if (wave == 0) split_factor_wave<0>(D, LD, cb, lane, ts);
  else if (wave == 1) split_factor_wave<1>(D, LD, cb, lane, ts);
  else if (wave == 2) split_factor_wave<2>(D, LD, cb, lane, ts);
  else if (wave == 3) split_factor_wave<3>(D, LD, cb, lane, ts);
}

}  // namespace cal
