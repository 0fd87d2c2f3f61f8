// bcr_kernels.hip — tree solver for the arrowhead normal equations on gfx950.
//
// Same system as solve_kernels.hip (what the reference hands to Ceres' DENSE_SCHUR, batch_optimizer.cpp:10-17,72-73):
//   [ B  E ] [y_s]   [g_s]      B: band over the spline control points (half bandwidth 6k-1),
//   [ Eᵀ C ] [y_c] = [g_c]      E: dense border (calibration columns), C: dense corner,
// with the Levenberg–Marquardt damping and Ceres' Jacobi scaling folded into the diagonal. The sequential band
// factorisation (band_cholesky_kernel: n_cp/2 dependent steps) is replaced by nested dissection in time:
//   * control points are grouped into superblocks of 5 (30 rows padded to 32); for spline order k <= 6 the band is
//     block TRIDIAGONAL in superblocks, bordered by the calibration columns and the right-hand side (F);
//   * level 0 eliminates chains of q consecutive superblocks between kept separators, every further level every other
//     surviving separator, all chains of a level side by side on different CUs (one launch per level); the last survivor
//     (the root) joins the calibration blocks in the dense reduced system, which reduced_solve_panel_kernel factors;
//   * eliminating superblock e with neighbours a (left separator) and n (next of the chain / right separator):
//       D_e = L Lᵀ,  [Z^A | Z^B | Z^F] = L⁻¹ [T(e,a) | T(n,e)ᵀ | F_e],
//       D_n -= Z^BᵀZ^B, T(n,a) -= Z^BᵀZ^A, F_n -= Z^BᵀZ^F, D_a -= Z^AᵀZ^A, F_a -= Z^AᵀZ^F, C -= Z^FᵀZ^F (one SYRK at the end);
//     a separator receives the contributions of the chains on either side in separate slots and adds them up in a
//     fixed order => bitwise reproducible, no atomics;
//   * back-substitution runs down the tree, y_e = L⁻ᵀ (L⁻¹g_e - Z^F y_c - Z^A y_a - Z^B y_n), and every node updates the
//     candidate point of its own control points.
// The dependent chain is (q + log2(n_cp / 5q)) block factorisations instead of n_cp / 2 band steps.
// The 32x32 factorisation (+ inverse, via 32 identity rows riding as extra rows) is two in-wave 16-column panels
// (panel_factor) and one MFMA tile update; everything else is v_mfma_f64_16x16x4_f64 tile products PᵀQ out of LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>

#include "problem_dev.hpp"
#include "solve_dev.hpp"
#include "block_elim.hpp"

namespace cal {

bool block_elim_enabled();
static int dense_elim_mode();

namespace {
constexpr int BP = kBcrBP;              // 32
constexpr int BB = BP * BP;             // 1024
constexpr int DLD = 33;                 // row stride of the augmented diagonal block [64][33]: rows 0..31 D, rows 32..63 identity -> L⁻ᵀ
constexpr int XLD = 2 * BP + kBcrFS + 1;  // row stride of X = [A | B | F slice] (80 columns)
constexpr int CA = 0, CB = BP, CF = 2 * BP;   // column offsets inside X / Z
constexpr int kLevelThreads = 512;
}  // namespace

// acc (+/-)= Σ_k P[k][pc0 + i] · Q[k][qc0 + j], k in [k0, k1): one 16x16 tile of PᵀQ on the matrix cores; P, Q row-major in LDS.
// Result layout: column j = lane & 15, row i = (lane >> 4) + 4·reg.
// All operands are read before the first MFMA is issued (one LDS round trip per tile, not one per k-step).
template <bool NEG, int NK>
DEVI f64x4 atb_tile_n(const double* P, int ldp, int pc0, const double* Q, int ldq, int qc0, f64x4 acc, int lane) {
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
template <bool NEG>
DEVI f64x4 atb_tile(const double* P, int ldp, int pc0, const double* Q, int ldq, int qc0, int k0, int k1, f64x4 acc, int lane) {
  // k0 is always 0 here; k1 is 16 (upper-triangular operand, first row tile) or 32
  return k1 - k0 == 16 ? atb_tile_n<NEG, 4>(P, ldp, pc0, Q, ldq, qc0, acc, lane) : atb_tile_n<NEG, 8>(P, ldp, pc0, Q, ldq, qc0, acc, lane);
}

// ---------------------------------------------------------------------------
// Entries of the damped system in superblock coordinates, straight from the reduce buffer R(x) (what prepare_kernel
// stages for the banded solver). Loads are unconditional (indices clamped), the structure is applied by selects.
// ---------------------------------------------------------------------------
struct FromR {
  const SolveArgs& a;
  const LmOptionsDev& o;
  double radius;
  // 0: the Jacobi scale comes from a.scale. 1 / 2: the first linear solve of a solve, whose bookkeeping (post_eval_body
  // with `first`: it computes a.scale from the diagonal of the first normal equations) rides in this very launch -- the
  // scale of a diagonal entry is formed from the entry itself, as that bookkeeping does (1: Jacobi scaling on, 2: off)
  int first_scale;
  DEVI double scale_of(double v_diag, int t) const {
    return first_scale == 0 ? a.scale[t] : (first_scale == 1 ? 1.0 / (1.0 + sqrt(v_diag)) : 1.0);
  }
  static constexpr int RB = 6 * kBcrCps;   // real rows of a superblock
  // tangent row of (superblock I, local row r); -1: padding, beyond the trajectory, or an unobserved control point
  DEVI int trow(int I, int r) const {
    const int t = RB * I + r;
    const bool ok = r < RB && t < a.n_s();
    return (ok && a.cp_active[(ok ? t : 0) / 6]) ? t : -1;
  }
  // H(tr, tc) inside the band, any order; 0 outside the band or when either index is -1
  DEVI double band(int tr, int tc) const {
    const int hi = max(tr, tc), lo = min(tr, tc);
    const int lo_c = max(lo, 0);
    const int ic = lo_c / 6, cc = lo_c % 6, ir = max(hi, 0) / 6, rr = max(hi, 0) % 6;
    const int d = ir - ic;
    const bool ok = lo >= 0 && d < a.k;
    const double v = a.R[a.off_B() + (size_t(ic) * a.k + (ok ? d : 0)) * 36 + cc * 6 + rr];
    return ok ? v : 0.0;
  }
  DEVI double damping(double v, int t) const {
    const double s = scale_of(v, t);
    return fmin(fmax(v * s * s, o.min_lm_diagonal), o.max_lm_diagonal) / (radius * s * s);
  }
  // D(I)(r, c), damped; identity on padding / unobserved rows. `file`: this thread owns dadd of the row.
  DEVI double diag_block(int I, int r, int c, bool file) const {
    const int tr = trow(I, r), tc = trow(I, c);
    double v = band(tr, tc);
    if (r == c) {
      const int t = RB * I + r;
      const double d = damping(v, max(tr, 0));
      if (tr < 0) { v = 1.0; if (file && r < RB && t < a.n_s()) a.dadd[t] = 0.0; }
      else { v += d; if (file) a.dadd[tr] = d; }
    }
    return v;
  }
  // H(superblock In row r, superblock Ic column c), In = Ic + 1
  DEVI double coupling(int In, int r, int Ic, int c) const { return band(trow(In, r), trow(Ic, c)); }
  // border row: calibration columns, right-hand side in column mc, zero padding
  DEVI double border(int I, int r, int j) const {
    const int tr = trow(I, r), t = max(tr, 0);
    const int jc = min(j, a.mc);
    const double e = a.R[j < a.mc ? a.off_E() + size_t(t) * a.mc + jc : a.off_g() + t];
    return (tr >= 0 && j <= a.mc) ? e : 0.0;
  }
};

// ---- hand-off inside a launch (the dense reduced solve -> the back-substitution workgroups riding in its launch) ----
// The producer's results leave with write-through (sc1) stores, are drained (s_waitcnt vmcnt(0)) and followed by an sc1
// flag store; the consumers poll the flag from one lane with relaxed sc1 loads and read the results with sc1 loads (they
// bypass the CU's vector L1, which another CU's stores never refresh). No agent-scope fence on either side: a release
// would write back the XCD's L2, an acquire invalidate the L1 -- microseconds each (MI355X_MICROARCH.md, "inter-workgroup
// visibility": the {sc1 stores, sc1 loads} form).
struct Handoff { int* word; int seq; };       // word == nullptr: plain kernel boundary, no hand-off
DEVI double load_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVI void store_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool HO> DEVI double load_y(const double* p) { return HO ? load_sc1(p) : *p; }
// all threads of the workgroup; the stores of every thread are drained before the flag goes up
DEVI void handoff_publish(const Handoff& h) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h.word, h.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// fan-in: every producing workgroup of a launch arrives once (its stores drained first), consumers wait for all of them
DEVI void fanin_arrive(int* word) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEVI void fanin_wait(const int* word, int n) {
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) __builtin_amdgcn_s_sleep(4);
  }
  __syncthreads();
}
DEVI void handoff_wait(const Handoff& h) {
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(h.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != h.seq) __builtin_amdgcn_s_sleep(4);
  }
  __syncthreads();
}

// One 16x16 tile (tr, tc) of the reduced system's calibration part, K-slice `slice` of `ks`: C(r, c) (+ damping on the
// diagonal, slice 0 only) - Σ_rows Y(row, r) Y(row, c) over the slice's rows of Y, on the matrix cores straight from
// global memory; NW waves split the rows and add up in a fixed order. With `late` (the tile rides in the last level's
// launch): the rows of the superblocks that level eliminates (late0, late1) are left out of the first pass -- they are
// being written by that very launch --, the workgroup then waits for the level's workgroups (fan-in word) and adds
// them, read with L1-bypassing loads.
template <int NW, bool LATE>
DEVI void schur_tile(const SolveArgs& a, const BcrArgs& b, const FromR& fr, int tile, int slice, int ks, double* sacc /* [NW][256] */,
                     const int* fan_word, int n_prod, int late0, int late1) {
  const int mc = a.mc, m = a.m, m1 = a.m + 1, m1p = b.m1p, m1y = a.mc + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t msq = size_t(m1) * m1;
  const int n = b.N * BP;                       // rows of Y (padding and root rows are zero)
  int tr = 0, rem = tile;
  while (rem > tr) { rem -= tr + 1; ++tr; }
  const int tc = rem;
  const int lc16 = lane & 15, lk = lane >> 4;
  const int rows_per = ((n + ks - 1) / ks + 15) & ~15;
  const int r_begin = slice * rows_per, r_end = min(n, r_begin + rows_per);
  const int per_wave = (((r_end - r_begin + NW - 1) / NW + 3) / 4) * 4;
  const int w_begin = r_begin + wave * per_wave, w_end = min(r_end, w_begin + per_wave);
  const int ca = 16 * tr + lc16, cb = 16 * tc + lc16;      // columns of Y: always < m1p
  const double* pa = b.Y + ca;
  const double* pb = b.Y + cb;
  const int f0 = LATE ? late0 : -1, f1 = LATE ? late1 : -1;
  // The entry of the damped corner this thread will store (threads 0..255: entry tid of the tile), requested FIRST: C(r, c) and
  // the diagonal's Jacobi scale are two dependent round trips (~2k clocks each) that depend on nothing this launch computes -- taken
  // behind the rows' sums (and, riding in the last level's launch, behind the fan-in) they were the tail of the launch.
  double s0_pre = 0.0;
  if (tid < 256) {
    const int r = tr * 16 + (tid >> 4), c = tc * 16 + (tid & 15);
    if (r < m1y && c <= r && slice == 0) {
      if (r < mc) {
        s0_pre = a.R[a.off_C() + size_t(r) * mc + c];
        if (r == c) { const double d = fr.damping(s0_pre, a.n_s() + r); a.dadd[a.n_s() + r] = d; s0_pre += d; }
      } else if (c < mc) s0_pre = a.R[a.off_g() + a.n_s() + c];
    }
  }
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  // (batches of 32 rows, FOUR of them requested before the first is consumed: a batch at a time the wave waited out a round trip
  //  per batch -- three in a row at configs[3] -- with nothing else to do)
  for (int kg = w_begin; kg < w_end; kg += 128) {
    double va[4][8], vb[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = kg + 32 * g + 4 * u + lk;
        const size_t ro = size_t(min(row, n - 1)) * m1p;
        va[g][u] = pa[ro]; vb[g][u] = pb[ro];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = kg + 32 * g + 4 * u + lk, sb = row >> 5;
        const bool use = row < w_end && sb != f0 && sb != f1;      // (selects on BOTH operands: a skipped row may hold anything, NaN included -- it is being rewritten by this very launch -- and 0 x NaN is NaN)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(use ? va[g][u] : 0.0, use ? vb[g][u] : 0.0, acc, 0, 0, 0);
      }
    }
  }
  if (LATE) {
    fanin_wait(fan_word, n_prod);
    // rows 32·blk + 4·wave + lk of the late superblocks (NW = 8 waves x 4 rows = 32), where they fall into this slice
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int sb = f == 0 ? f0 : f1;
      const int row = BP * max(sb, 0) + 4 * wave + lk;
      const bool use = sb >= 0 && wave < 8 && row >= r_begin && row < r_end;
      const size_t ro = size_t(min(row, n - 1)) * m1p;
      const double va = load_sc1(pa + ro), vb = load_sc1(pb + ro);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(use ? va : 0.0, use ? vb : 0.0, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sacc[wave * 256 + (lk + 4 * r) * 16 + lc16] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int ti = tid >> 4, tj = tid & 15;
    const int r = tr * 16 + ti, c = tc * 16 + tj;      // columns of Y
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += sacc[w * 256 + tid];
    if (r < m1y && c <= r) {
      const int fi = r < mc ? r : m, fc = c < mc ? c : m;   // final index: the right-hand side sits behind the root rows
      // corner of the damped system: C(r, c) (+ damping on the diagonal), right-hand side g_c in row m; (m, m) is unused
      a.Spart[size_t(slice) * msq + size_t(fi) * m1 + fc] = s0_pre - sum;
    }
  }
}
// Root rows of the reduced system: Spart(mc + r, j) = F_root(r, j), Spart(mc + r, mc + r2) = D_root(r, r2), Spart(m, mc + r) =
// F_root(r, mc), each plus the root's pending slots; workgroup w0 of nw, NT threads. SC1: read behind an in-launch fan-in.
template <bool SC1>
DEVI void schur_root_rows(const SolveArgs& a, const BcrArgs& b, int ks, int w0, int nw, int NT) {
  if (b.root < 0) return;
  const int mc = a.mc, m = a.m, m1 = a.m + 1, m1p = b.m1p, m1y = a.mc + 1;
  const size_t msq = size_t(m1) * m1;
  const int tid = threadIdx.x;
  const int br = m - mc;
  const size_t fblk = size_t(BP) * m1p;
  const double* pD = b.pendD + size_t(b.root_par) * size_t(b.N) * 2 * BB;
  const double* pF = b.pendF + size_t(b.root_par) * size_t(b.N) * 2 * fblk;
  const int mask = b.root_pend;
  const int total = br * (m1y + br);
  auto ld = [](const double* p) { return SC1 ? load_sc1(p) : *p; };
  for (int e = w0 * NT + tid; e < total; e += nw * NT) {
    const int r = e / (m1y + br), j = e % (m1y + br);
    const bool in_f = j < m1y;
    const int r2 = j - m1y;
    if (!in_f && r2 > r) continue;
    // the three sources of an entry are requested together (clamped, unconditional): one round trip per entry, not three
    const size_t o = in_f ? size_t(r) * m1p + j : size_t(r) * BP + r2;
    const double* base = in_f ? b.F + size_t(b.root) * fblk : b.D + size_t(b.root) * BB;
    const double* p0 = in_f ? pF + (size_t(b.root) * 2 + 0) * fblk : pD + (size_t(b.root) * 2 + 0) * BB;
    const double* p1 = in_f ? pF + (size_t(b.root) * 2 + 1) * fblk : pD + (size_t(b.root) * 2 + 1) * BB;
    const double v0 = ld(base + o), v1 = ld(p0 + o), v2 = ld(p1 + o);
    const double v = (v0 + ((mask & 1) ? v1 : 0.0)) + ((mask & 2) ? v2 : 0.0);
    int fi, fc;
    if (in_f) { if (j < mc) { fi = mc + r; fc = j; } else { fi = m; fc = mc + r; } }
    else { fi = mc + r; fc = mc + r2; }
    a.Spart[size_t(fi) * m1 + fc] = v;
    for (int k = 1; k < ks; ++k) a.Spart[size_t(k) * msq + size_t(fi) * m1 + fc] = 0.0;
  }
}

// ---------------------------------------------------------------------------
// One level of the elimination tree. Workgroup (node, role): role 0 owns the spine (it files L⁻ᵀ, Z^A, Z^B, the
// separators' diagonal updates and the fill between them), role s >= 1 the border columns [16(s-1), 16s) (Z^F and the
// separators' border updates). Every role repeats the chain's factorisations (same instructions, same inputs =>
// identical factors), which costs no time and no communication. Extra workgroups behind the nodes add last level's
// pending updates to the separators that survive this level, too.
// ---------------------------------------------------------------------------
// FROM_R (level 0): the chain blocks come straight from the reduce buffer R(x) with the damping applied on the fly (no
// staging pass); the extra workgroups initialise D and F of the level's separators from R, and the last of them does
// the bookkeeping of the step just accepted when `with_post` is set (post_eval_body: it only reads R(x) and x).
// ELIM (round 4): the block is eliminated by block_elim.hpp -- a chief wave on the spine, waves 1..3 as followers with the
// identity rows (role 0: L⁻ᵀ) and the rows of Xᵀ, so that Z comes out of the factorisation -- instead of panel | tile |
// panel | Z = MᵀX (CALICO_ELIM=panel keeps those).
// A thread's share of a block's entries on the way from memory to LDS (bcr_level_kernel's fetch / commit): NL threads
// take NU entries of D / B / A and NF of the F slice each, thread lt the entries lt, lt + NL, ...
// Level 0 (round 5): the entries are dealt so that the lanes of a load run ALONG the six contiguous doubles of the band's
// storage (R holds H(row, column) at [(column's control point, distance)][column component][row component]: the six row
// components of one column are contiguous) -- D: only the upper triangle r <= c, row-major (528 entries, mirrored into LDS:
// the block is symmetric; the lower half read row-major walked R with a stride of six doubles), B / A: entry e is (column
// e >> 5, row e & 31). A 64-lane request touches ~11 lines instead of up to 64; the loaders' request -> commit time is what
// ends every step of a chain and the head of the launch.
constexpr int kBcrTri = BP * (BP + 1) / 2;       // 528
template <int NL_, int NU_, int NF_>
struct BcrLoadMap {
  static constexpr int NL = NL_, NU = NU_, NF = NF_;
  static constexpr int NUD = (kBcrTri + NL_ - 1) / NL_;      // level 0: entries of D's upper triangle per thread
  // iD / rcD / okD are declared [NU] and walked up to NUD; fetch_request packs the entries' validity bits of D, B, A and F
  // into one byte each of a 32-bit word
  static_assert(NUD <= NU_ && NU_ <= 8 && NF_ <= 8, "BcrLoadMap: per-thread entry counts must fit the arrays and the flag bytes");
  int lt;
  int iD[NU_], iB[NU_];        // level 0: positions in R of the entries for superblock 0
  int rcD[NU_];                // level 0: (r << 8 | c), r <= c, of the thread's D entries
  bool okD[NU_], okB[NU_];
};
template <int NU, int NF>
struct BcrPre { double d[NU], bt[NU], at[NU], f[NF]; };
// level 0: what a block's requests return, before selection and damping (fetch_request / fetch_finish)
template <int NUD, int NU, int NF>
struct BcrRaw { double vraw[NUD], svv[NUD], q2v[NUD], vraw1[NUD], graw[NU], garaw[NU], graw1[NU], garaw1[NU], fraw[NF], fraw1[NF]; unsigned flags; };
#ifndef BCR_FIRST_BLOCK_ALL_WAVES
#define BCR_FIRST_BLOCK_ALL_WAVES 1
#endif
// BCR_EARLY_REQUESTS=1 (compile time; round 5, measured and left OFF): the loaders issue the requests of block i + 2 in front of
// step i's Schur phase and only finish (selection, damping, commit) at the top of step i + 1. Bit-identical; level 0 25.0 ->
// 26.9 us (27.8 with the 18 registers it spills): the requests' address arithmetic (3-4k clocks on SIMDs shared with the chief
// and the followers) then sits between the step's two barriers and holds the chief up by more than the earlier commit gains.
#ifndef BCR_EARLY_REQUESTS
#define BCR_EARLY_REQUESTS 0
#endif

// LA (look-ahead, round 5; ELIM only): between two blocks of a chain only what the NEXT block's chief waits for stays in
// front of the step's second barrier -- the operands of every Schur product read into registers and the three tiles of
// next.D -= Z^BᵀZ^B --; the chief then starts on the next block at once. The followers of the A and F tiles form the update
// of their own input tiles (next.A -= Z^BᵀZ^A, next.F -= Z^BᵀZ^F) straight in the registers the elimination keeps them in
// (elim_follow with use_pre: same layout), the loader waves file Z and carry the left separator's sums, and the next block's
// requests go out right behind the barrier. Same products in the same order as without it: bit-identical.
// Rolling chief (bcr_level_kernel<.., ROLL>): where a lane's tile entries sit in the band's storage, by spline order k = 1..6 --
// the band is uniform in time, so the positions are those of superblock 0 (superblock I adds I·strideB) and depend on nothing
// but k and the lane. Per lane 48 words: [0..11] byte offsets of the spine's entries (tiles (0,0), (0,1), (1,1), register 0..3 each:
// elim_load_spine), [12..27] of the rows of Bᵀ (tile qt, column half h, register r at 12 + (2 qt + h) 4 + r: elim_load_rows), [28] which
// of them exist (spine: bits 0..11, Bᵀ: bits 16..31), [29] the same for Aᵀ, [32..47] byte offsets of the rows of Aᵀ (the first block
// of a chain against the separator on its left, counted from THAT superblock's storage). Computed on the host at configure time:
// in the kernel it was ~450 instructions of index arithmetic per wave in front of the first request -- cold code at the head of a launch.
constexpr int kRollTabWords = 48;
__device__ unsigned g_roll_tab[6][64][kRollTabWords];
static void fill_roll_table(unsigned (*tab)[64][kRollTabWords]) {
  constexpr int RB = 6 * kBcrCps;
  for (int k = 1; k <= 6; ++k) {
    auto pos = [&](int hi, int lo, bool& ok) {      // H(hi, lo), lo <= hi, counted from the start of lo's superblock
      const int d = hi / 6 - lo / 6;
      ok = d < k;
      return unsigned(8 * (((lo / 6) * k + (ok ? d : 0)) * 36 + (lo % 6) * 6 + hi % 6));
    };
    for (int lane = 0; lane < 64; ++lane) {
      unsigned* t = tab[k - 1][lane];
      for (int i = 0; i < kRollTabWords; ++i) t[i] = 0;
      const int l16 = lane & 15, lk = lane >> 4;
      unsigned okS = 0, okB = 0, okA = 0;
      for (int r = 0; r < 4; ++r) {
        const int c = lk + 4 * r, hi = std::max(l16, c), lo = std::min(l16, c);
        bool ok;
        t[r] = pos(hi, lo, ok); if (ok) okS |= 1u << r;
        t[4 + r] = pos(16 + l16, c, ok); if (ok && 16 + l16 < RB) okS |= 1u << (4 + r);
        t[8 + r] = pos(16 + hi, 16 + lo, ok); if (ok && 16 + hi < RB) okS |= 1u << (8 + r);
      }
      for (int qt = 0; qt < 2; ++qt)
        for (int h = 0; h < 2; ++h)
          for (int r = 0; r < 4; ++r) {
            const int e = (qt * 2 + h) * 4 + r;
            bool ok;
            {      // Bᵀ: the next block's row rn against this block's column c
              const int c = 16 * h + lk + 4 * r, rn = 16 * qt + l16;
              t[12 + e] = pos(RB + rn, c, ok);
              if (ok && rn < RB && c < RB) okB |= 1u << e;
            }
            {      // Aᵀ: this block's row rb against the left separator's column cs
              const int rb = 16 * h + lk + 4 * r, cs = 16 * qt + l16;
              t[32 + e] = pos(RB + rb, cs, ok);
              if (ok && rb < RB && cs < RB) okA |= 1u << e;
            }
          }
      t[28] = okS | okB << 16;
      t[29] = okA;
    }
  }
}
// (test hook, host only: the row of the table for spline order k and a lane -- tests/test_host_abi.py checks it against the band's layout)
void roll_table_row(int k, int lane, unsigned* out) {
  static unsigned tab[6][64][kRollTabWords];
  static std::once_flag once;
  std::call_once(once, [] { fill_roll_table(tab); });
  for (int i = 0; i < kRollTabWords; ++i) out[i] = tab[k - 1][lane][i];
}
static hipError_t upload_roll_table() {      // (called under configure_kernels' lock)
  static unsigned host_tab[6][64][kRollTabWords];
  static std::once_flag once;
  std::call_once(once, [] { fill_roll_table(host_tab); });
  static bool done[64] = {};       // once per device: the symbol lives on each of them, and the copy waits for the device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_roll_tab), host_tab, sizeof(host_tab));
  if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
  return e;
}

// ROLL (round 6; level 0 with the block elimination): the chain without a workgroup barrier between its blocks -- see the
// "rolling chief" section in front of the step loop.
template <bool FROM_R, bool ELIM, bool LA = false, bool ROLL = false>
__global__ __launch_bounds__(kLevelThreads) void bcr_level_kernel(SolveArgs a, BcrArgs b, int node0, int n_nodes, int nfs, int level,
                                                                   int keep0, int n_keep, LmOptionsDev o, int with_post,
                                                                   const double* __restrict__ x, const BlockDev* __restrict__ blocks,
                                                                   int n_blocks, IterLog* log, int log_cap, int jacobi_scaling,
                                                                   int n_schur_wg, int n_root_wg, int schur_ks, int* fan_word, int n_prod,
                                                                   BcrInlineNodes inl) {
  const long long t_kernel = CAL_DEV_TIMING(a.debug != 0) ? __builtin_readcyclecounter() : 0;
  LmState* st = a.st;
  // (vector loads: see vector_ptr. `terminated` is tested after the first loads are on their way -- they are harmless)
  const LmState* const stv = vector_ptr(const_cast<const LmState*>(st));
  const int terminated_v = stv->terminated;
  const double radius = stv->radius;          // (stays in a VGPR: only arithmetic uses it)
  const int r_cur_v = FROM_R ? stv->rcur : 0;
  // ROLL: the lanes' offset table (g_roll_tab) is asked for with the state, at the kernel's first instructions -- its round trip
  // (~4k clocks behind a kernel boundary) then runs beside the state's and the set-up below instead of behind them. Every wave
  // asks for the same words, used or not: straight-line requests, nothing waits for them before they are used.
  uint4 tv[8], tva[4];
  if constexpr (ROLL && FROM_R) {
    const uint4* const tl = reinterpret_cast<const uint4*>(&g_roll_tab[min(max(a.k, 1), 6) - 1][threadIdx.x & 63][0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) tv[i] = tl[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) tva[i] = tl[8 + i];
  }
  // This kernel's code asked for as data, so that it stands in the XCD's L2 when the instruction fetch comes for it (see
  // prefetch_code): behind a kernel boundary the code of a launch is not in the L2 any more -- the evaluation and the gather move
  // 40 MB through it per iteration --, and a wave that runs into a line nobody has fetched waits ~2k clocks for it.
  // Not in the rolling form: there the head IS the critical path (table and state -> requests -> first tiles), and these ~300
  // misses per workgroup in front of them -- by all waves, or by waves 4..6 alone -- made level 0 1.0 / 1.35 us LONGER
  // (r06_code_prefetch_ab.txt); the barrier form waits for its first block at a barrier anyway: level 1 11.0 -> 10.5 us.
  int code_pf = 0;
  if constexpr (!ROLL) code_pf = prefetch_code(threadIdx.x, kLevelThreads, 40 * 1024);
  // `pub` (the last level's launch when the Schur complement rides in it): this level's workgroups are the PRODUCERS of
  // an in-launch fan-in -- what the riders read (Y rows, the root's pending slots, separators updated in place) leaves
  // with write-through stores, and every producing workgroup arrives at `fan_word` once, terminated or not; the riders
  // behind them in the grid are the Schur complement's tiles and the root's rows of the reduced system.
  const bool pub = !FROM_R && fan_word != nullptr;
  if (FROM_R && fan_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *fan_word = 0;     // (the next fan-in counts from zero)
  if (pub && int(blockIdx.x) >= int(gridDim.x) - n_schur_wg - n_root_wg) {
    if (uniform(terminated_v)) return;
    use_current_R(a);
    const FromR frs = {a, o, radius, 0};
    extern __shared__ double lds_s[];
    const int w = int(blockIdx.x) - (int(gridDim.x) - n_schur_wg - n_root_wg);
    if (w < n_schur_wg) {
      // (the last level has one or two nodes of one superblock each)
      const int lb0 = inl.n > 0 ? inl.nd[0].blk0 : b.nodes[node0].blk0, lb1 = n_nodes > 1 ? (inl.n > 0 ? inl.nd[1].blk0 : b.nodes[node0 + 1].blk0) : -1;
      schur_tile<kLevelThreads / 64, true>(a, b, frs, w / schur_ks, w % schur_ks, schur_ks, lds_s, fan_word, n_prod, lb0, lb1);
      if (CAL_DEV_TIMING((a.debug == 4 || a.debug == 6) && threadIdx.x == 0 && w < 2)) printf("bcr_level %d: Schur tile rider %d lived %lld clocks\n", level, w, (long long)(__builtin_readcyclecounter() - t_kernel));
    } else {
      fanin_wait(fan_word, n_prod);
      const long long t_fan = CAL_DEV_TIMING(a.debug == 4 || a.debug == 6) ? __builtin_readcyclecounter() - t_kernel : 0;
      schur_root_rows<true>(a, b, schur_ks, w - n_schur_wg, n_root_wg, kLevelThreads);
      if (CAL_DEV_TIMING((a.debug == 4 || a.debug == 6) && threadIdx.x == 0 && w == n_schur_wg)) printf("bcr_level %d: root-row rider: fan-in complete at %lld clocks, lived %lld\n", level, t_fan, (long long)(__builtin_readcyclecounter() - t_kernel));
    }
    return;
  }
  auto put = [&](double* dst, double v) { if (pub) store_sc1(dst, v); else *dst = v; };
  if (FROM_R) {
    if (with_post && blockIdx.x == gridDim.x - 1) {
      if (uniform(terminated_v)) return;
      if (threadIdx.x == 0 && a.st->commit_pending) a.st->commit_pending = 0;   // see commit_kernel (several ranks)
      post_eval_body(a, x, blocks, n_blocks, o, log, log_cap, with_post == 2 ? 1 : 0, jacobi_scaling);     // (all threads: it has barriers inside)
      if (CAL_DEV_TIMING(a.debug == 4 && threadIdx.x == 0)) printf("bcr_level 0: the bookkeeping workgroup lived %lld clocks\n", (long long)(__builtin_readcyclecounter() - t_kernel));
      return;
    }
  }
  // Which reduce buffer holds R(x) is a word of the LM state (speculative evaluation): the chains' first block is requested
  // from BOTH buffers at once and selected when the state has arrived -- with the buffer chosen first, every load of the
  // launch's head waited out the state's round trip before it was even issued (two dependent round trips instead of one).
  const double* const R_buf0 = a.R;
  const double* const R_buf1 = a.R + a.r_stride;
  const FromR fr = {a, o, radius, (FROM_R && with_post == 2) ? (jacobi_scaling ? 1 : 2) : 0};
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform for the compiler too: the roles of the waves are alternatives -- scalar branches, registers per path -- not masked regions that run one after the other)
  // Workgroup -> (node, role), XCD-aware: workgroup p runs on XCD p % 8, each XCD has its own L2, and all roles of a node
  // read the same spine blocks -- so node c takes the workgroups p = (c % 8) + 8·slot: one L2 serves its roles. The
  // first 8·slots workgroups are (node, role) pairs (those of nodes that do not exist return), the others are the
  // extra workgroups below.
  const int per = 1 + nfs;
  const int main_span = 8 * ((n_nodes + 7) / 8) * per;
  const int pcol = int(blockIdx.x) & 7, pslot = int(blockIdx.x) >> 3;
  const int node_l = (pslot / per) * 8 + pcol;
  const bool extra_wg = int(blockIdx.x) >= main_span;
  if (!extra_wg && node_l >= n_nodes) return;
  const int bid = extra_wg ? n_nodes * per + (int(blockIdx.x) - main_span) : node_l * per + pslot % per;
  const int grid_l = n_nodes * per + (int(gridDim.x) - n_schur_wg - n_root_wg - main_span);      // logical grid: nodes × roles, then the extras
  const int m1p = b.m1p, par = level & 1;
  const size_t NB = size_t(b.N);
  const size_t fblk = size_t(BP) * m1p;
  double* const pendD_w = b.pendD + size_t(par) * NB * 2 * BB;
  double* const pendF_w = b.pendF + size_t(par) * NB * 2 * fblk;
  const double* const pendD_r = b.pendD + size_t(par ^ 1) * NB * 2 * BB;
  const double* const pendF_r = b.pendF + size_t(par ^ 1) * NB * 2 * fblk;
  const double* const Gr = b.G + size_t(par) * NB * BB;
  double* const Gw = b.G + size_t(par ^ 1) * NB * BB;
  if (bid >= n_nodes * per) {
    if (uniform(terminated_v)) { if (pub) fanin_arrive(fan_word); return; }
    if (FROM_R) use_current_R(a);
    // surviving separators that are not eliminated at this level: D += pending, F += pending (in place; nobody else
    // reads them in this launch). At level 0 they are initialised from R(x) instead.
    const size_t aw = size_t(bid - n_nodes * per), naw = size_t(grid_l - n_nodes * per - (FROM_R && with_post ? 1 : 0));
    const size_t per_blk = size_t(BB) + fblk;
    if (FROM_R) {
      // four entries per thread and pass, their reads of R (and of the Jacobi scale, for the diagonal) requested together:
      // one entry at a time the loop waited out two dependent round trips per entry, and with a long trajectory's sixty
      // separators these workgroups, not the chains, ended the launch (1453 control points: level 0 45 us)
      const size_t total = size_t(n_keep) * per_blk, stride = naw * kLevelThreads;
      const int n_s = a.n_s();
      for (size_t e0 = aw * kLevelThreads + tid; e0 < total; e0 += 4 * stride) {
        size_t ridx[4], didx[4];
        int trd[4], flags[4];          // flags: 1 entry exists, 2 value comes from R, 4 diagonal entry, 8 row exists in the trajectory, 16 goes to b.F
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t e = e0 + u * stride;
          const size_t ec = e < total ? e : total - 1;
          const int kb = int(ec / per_blk);
          const size_t rem = ec - size_t(kb) * per_blk;
          const int blk = b.keep[2 * (keep0 + kb)];
          int fl = e < total ? 1 : 0;
          size_t ri = 0;
          int tr_d = 0;
          if (rem < size_t(BB)) {
            const int r = int(rem) >> 5, c = int(rem) & 31;
            const int tr = fr.trow(blk, r), tc = fr.trow(blk, c);
            const int hi = max(tr, tc), lo = min(tr, tc);
            const int lo_c = max(lo, 0), ic = lo_c / 6, cc = lo_c % 6, ir = max(hi, 0) / 6, rr = max(hi, 0) % 6, d = ir - ic;
            const bool ok = lo >= 0 && d < a.k;
            ri = a.off_B() + (size_t(ic) * a.k + (ok ? d : 0)) * 36 + cc * 6 + rr;
            if (ok) fl |= 2;
            if (r == c) { fl |= 4; const int t = FromR::RB * blk + r; if (r < FromR::RB && t < n_s) fl |= 8; tr_d = tr >= 0 ? tr : -1 - (t < n_s ? t : 0); }
            didx[u] = size_t(blk) * BB + rem;
          } else {
            const size_t q = rem - BB;
            const int r = int(q / m1p), j = int(q % m1p);
            const int tr = fr.trow(blk, r), t = max(tr, 0), jc = min(j, a.mc);
            ri = j < a.mc ? a.off_E() + size_t(t) * a.mc + jc : a.off_g() + t;
            if (tr >= 0 && j <= a.mc) fl |= 2;
            fl |= 16;
            didx[u] = size_t(blk) * fblk + q;
          }
          ridx[u] = ri; trd[u] = tr_d; flags[u] = fl;
        }
        double rv[4], sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { rv[u] = a.R[ridx[u]]; sc[u] = a.scale[(flags[u] & 4) ? max(trd[u], 0) : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int fl = flags[u];
          if (!(fl & 1)) continue;
          double v = (fl & 2) ? rv[u] : 0.0;
          if (fl & 4) {        // diagonal: LM damping of an observed row, identity for padding / unobserved rows (see FromR::diag_block)
            if (trd[u] >= 0) {
              const double sv = fr.first_scale == 0 ? sc[u] : fr.scale_of(v, 0);
              const double dd = fmin(fmax(v * sv * sv, o.min_lm_diagonal), o.max_lm_diagonal) / (radius * sv * sv);     // FromR::damping
              v += dd; a.dadd[trd[u]] = dd;
            } else {
              v = 1.0;
              if (fl & 8) a.dadd[-1 - trd[u]] = 0.0;
            }
          }
          if (fl & 16) b.F[didx[u]] = v; else b.D[didx[u]] = v;
        }
      }
      if (CAL_DEV_TIMING(a.debug == 4 && threadIdx.x == 0 && aw == 0)) printf("bcr_level 0: a separator-initialisation workgroup lived %lld clocks\n", (long long)(__builtin_readcyclecounter() - t_kernel));
      if (pub) fanin_arrive(fan_word);
      return;
    }
    for (size_t e = aw * kLevelThreads + tid; e < size_t(n_keep) * per_blk; e += naw * kLevelThreads) {
      const int kb = int(e / per_blk);
      const size_t rem = e % per_blk;
      const int blk = b.keep[2 * (keep0 + kb)], mask = b.keep[2 * (keep0 + kb) + 1];
      if (rem < size_t(BB)) {
        double v = b.D[size_t(blk) * BB + rem];
        if (mask & 1) v += pendD_r[(size_t(blk) * 2 + 0) * BB + rem];
        if (mask & 2) v += pendD_r[(size_t(blk) * 2 + 1) * BB + rem];
        put(b.D + size_t(blk) * BB + rem, v);
      } else {
        const size_t q = rem - BB;
        double v = b.F[size_t(blk) * fblk + q];
        if (mask & 1) v += pendF_r[(size_t(blk) * 2 + 0) * fblk + q];
        if (mask & 2) v += pendF_r[(size_t(blk) * 2 + 1) * fblk + q];
        put(b.F + size_t(blk) * fblk + q, v);
      }
    }
    if (pub) fanin_arrive(fan_word);
    return;
  }
  // The node's descriptor without a load in front of the node's requests (a dependent round trip behind a kernel boundary is
  // ~1.9 us): level 0 is regular -- [chain of q] [separator] [chain] ... in time order --, so its descriptors are arithmetic on
  // the node's number (inl.q_regular; the host's table says the same); the few nodes of the levels near the top come with
  // the kernel arguments (inl.n); only the wide middle levels of a long trajectory read the table.
  BcrNodeDev nd_;
  {
    const int c = bid / per;
    if (FROM_R && inl.q_regular > 0) {
      nd_.blk0 = c * (inl.q_regular + 1); nd_.q = min(inl.q_regular, b.N - nd_.blk0);
      nd_.left = c > 0 ? nd_.blk0 - 1 : -1; nd_.right = nd_.blk0 + nd_.q < b.N ? nd_.blk0 + nd_.q : -1; nd_.slot = node0 + c; nd_.pend = 0;
    } else if (!FROM_R && inl.n > 0) {
      nd_ = inl.nd[0];
      if (c == 1) nd_ = inl.nd[1];
      if (c == 2) nd_ = inl.nd[2];
      if (c == 3) nd_ = inl.nd[3];
    } else nd_ = b.nodes[node0 + c];
  }
  const int role = bid % per;
  const int f0 = (role - 1) * kBcrFS;        // first border column of this role (role >= 1)
  const int q = nd_.q, left = nd_.left, right = nd_.right, blk0 = nd_.blk0, pend_mask = nd_.pend;
  extern __shared__ double lds[];
  double* const Daug = lds;                        // [2][64·DLD]
  double* const Xb = Daug + 2 * 64 * DLD;          // [2][32·XLD]
  double* const Zb_base = Xb + 2 * BP * XLD;       // [32·XLD], LA: [2][32·XLD]
  double* const dinv = Zb_base + (LA ? 2 : 1) * BP * XLD;    // [80]
  double* const bcast = dinv + 80;                 // [128]
  double* const dump = bcast + 128 + tid;          // [512]
  const ElimChannel ech = elim_channel(bcast + 128 + kLevelThreads);      // [kElimBufDoubles] (ELIM)
  static_assert(!ROLL || (ELIM && !LA), "the rolling chief works on the block elimination");
  if (ELIM && !ROLL) elim_reset(ech, tid, kLevelThreads);   // (the barrier behind the first block's commit orders it)
  const int l16 = lane & 15, lk = lane >> 4;
  // ---- global -> registers -> LDS of one chain block. Loaders are waves 1-3 and 5-7 (384 threads: three entries each
  // of D / B / A, two of the F slice); wave 0 goes straight to the factorisation, it is the critical path of every step. ----
  // (ELIM: waves 1..3 are the chief's followers from the first clock of a step -- a follower that issues loads first starts
  //  late and ends the step late --, so the loaders are waves 4..7: four entries each of D / B / A, two of the F slice.
  //  Waves 5..7 alone take 8.3k clocks from request to commit, longer than the chief's chain.)
  constexpr int NL = ELIM ? 256 : kLevelThreads - 128, NU = ELIM ? 4 : 3, NF = 2;
  const bool loader = ELIM ? wave >= 4 : (wave != 0 && wave != 4);    // wave 4 shares the panel wave's SIMD: it stays out of the way, too
  const int lt = ELIM ? tid - 256 : tid - 64 - (wave > 4 ? 64 : 0);
  typedef BcrPre<NU, NF> Pre;
  // level 0: positions in R of this thread's entries for superblock 0 (the band is uniform in time: superblock I adds
  // I·stride), and what does not depend on the superblock of their validity
  constexpr int RB = 6 * kBcrCps;
  const int strideB = kBcrCps * a.k * 36;
  auto fill_map = [&](auto& m) {
    typedef typename std::remove_reference<decltype(m)>::type M;
    if (FROM_R) {
#pragma unroll
      for (int u = 0; u < M::NUD; ++u) {      // D: entry eD of the row-major upper triangle -> (r, c), r <= c
        const int eD = min(max(m.lt, 0) + M::NL * u, kBcrTri - 1);
        // row r starts at r (2 BP + 1 - r) / 2: the float estimate is exact to +-1, fixed up
        int r = int((float(2 * BP + 1) - sqrtf(float((2 * BP + 1) * (2 * BP + 1)) - 8.0f * float(eD))) * 0.5f);
        r = max(0, min(BP - 1, r));
        if (((r + 1) * (2 * BP - r)) / 2 <= eD) ++r;
        if ((r * (2 * BP + 1 - r)) / 2 > eD) --r;
        const int c = r + eD - (r * (2 * BP + 1 - r)) / 2;
        m.rcD[u] = r << 8 | c;
        const int dD = c / 6 - r / 6;
        m.okD[u] = r < RB && c < RB && dD < a.k;
        m.iD[u] = m.okD[u] ? ((r / 6) * a.k + dD) * 36 + (r % 6) * 6 + c % 6 : 0;
      }
#pragma unroll
      for (int u = 0; u < M::NU; ++u) {       // B / A: entry e -> (column e >> 5, row e & 31): lanes along the rows
        const int e = min(max(m.lt, 0) + M::NL * u, BB - 1);
        const int c = e >> 5, r = e & 31;
        const int dB = kBcrCps + r / 6 - c / 6;       // row r of the next superblock against column c of this one
        m.okB[u] = r < RB && c < RB && dB < a.k;
        m.iB[u] = m.okB[u] ? ((c / 6) * a.k + dB) * 36 + (c % 6) * 6 + r % 6 : 0;
      }
    }
  };
  BcrLoadMap<NL, NU, NF> lmap;
  lmap.lt = lt;
  constexpr bool kAllFetchFirst = ELIM && BCR_FIRST_BLOCK_ALL_WAVES;
  if (!kAllFetchFirst) fill_map(lmap);       // (otherwise behind the first block's requests: the loaders' map is for the later blocks)
  // Level 0: a block's fetch in two halves -- `fetch_request` is the address arithmetic and the loads (nothing in it waits),
  // `fetch_finish` the selection, the damping of the diagonal and what goes to `pr` -- so that the loaders can put a block's
  // requests in FRONT of a step's Schur phase and only finish behind it (round 5; see the step loop).
  auto fetch_request = [&](int i, auto& rw, auto both_tag, const auto& m) {
    typedef typename std::remove_reference<decltype(m)>::type M;
    constexpr int NL = M::NL, NU = M::NU, NF = M::NF, NUD = M::NUD;
    const int lt = m.lt;
    constexpr bool BOTH = decltype(both_tag)::value;      // (first block of the launch: see R_buf0 / R_buf1)
    const int blk = blk0 + i;
    const bool has_next = (i + 1 < q) || right >= 0;
    const bool has_a = (i == 0) && left >= 0;
    // neighbours in the tree are neighbours in time at level 0 (next = blk + 1, left separator = blk - 1)
    const int nreal = a.n_s() - RB * blk, nreal_n = nreal - RB;     // real rows of this superblock / of the next one
    const double* const Rsel = BOTH ? R_buf0 : a.R;
    const double* RBnd = Rsel + a.off_B() + size_t(blk) * strideB;
    const double* RBndA = Rsel + a.off_B() + size_t(max(blk - 1, 0)) * strideB;
    const size_t alt = BOTH ? a.r_stride : 0;      // the same entry of the other buffer
    unsigned fl = 0;
#pragma unroll
    for (int u = 0; u < NUD; ++u) {        // D, upper triangle
      const int r = m.rcD[u] >> 8, c = m.rcD[u] & 255;
      bool act_r = true, act_c = true;
      if (!b.all_active) {
        const int n_cp = a.n_cp;
        act_r = a.cp_active[min(kBcrCps * blk + r / 6, n_cp - 1)] != 0;
        act_c = a.cp_active[min(kBcrCps * blk + c / 6, n_cp - 1)] != 0;
      }
      const bool vD = m.okD[u] && r < nreal && c < nreal && act_r && act_c;
      fl |= vD ? 1u << u : 0u;
      rw.vraw[u] = RBnd[vD ? m.iD[u] : 0];
      if (BOTH) rw.vraw1[u] = RBnd[alt + (vD ? m.iD[u] : 0)];
      const int ts = (r == c && r < RB && r < nreal) ? RB * blk + r : 0;
      rw.svv[u] = a.scale[ts]; rw.q2v[u] = a.scale[a.NT() + ts];        // (harmless during a solve's first linear solve, which does not use them)
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {         // B (next superblock's rows against this one's columns), A (this one's rows against the left separator's columns)
      const int e = min(max(lt, 0) + NL * u, BB - 1);
      const int c = e >> 5, r = e & 31;
      bool act_r = true, act_c = true, act_rn = true, act_cl = true;
      if (!b.all_active) {
        const int n_cp = a.n_cp;
        act_r = a.cp_active[min(kBcrCps * blk + r / 6, n_cp - 1)] != 0;
        act_c = a.cp_active[min(kBcrCps * blk + c / 6, n_cp - 1)] != 0;
        act_rn = a.cp_active[min(kBcrCps * (blk + 1) + r / 6, n_cp - 1)] != 0;
        act_cl = a.cp_active[min(max(kBcrCps * (blk - 1) + c / 6, 0), n_cp - 1)] != 0;
      }
      const bool vB = m.okB[u] && has_next && r < nreal_n && act_rn && act_c;
      fl |= vB ? 1u << (8 + u) : 0u;
      rw.graw[u] = RBnd[vB ? m.iB[u] : 0];
      if (BOTH) rw.graw1[u] = RBnd[alt + (vB ? m.iB[u] : 0)];
      const bool vA = m.okB[u] && has_a && r < nreal && act_r && act_cl;
      fl |= vA ? 1u << (16 + u) : 0u;
      rw.garaw[u] = 0.0;
      if (has_a) rw.garaw[u] = RBndA[vA ? m.iB[u] : 0];      // (only the first block of a chain touches the left separator)
      rw.garaw1[u] = 0.0;
      if (BOTH && has_a) rw.garaw1[u] = RBndA[alt + (vA ? m.iB[u] : 0)];
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = min(max(lt, 0) + NL * u, BP * kBcrFS - 1);
      const int r = e >> 4, col = role > 0 ? f0 + (e & 15) : 0;
      const int t = RB * blk + r;
      bool act_r = true;
      if (!b.all_active) act_r = a.cp_active[min(kBcrCps * blk + r / 6, a.n_cp - 1)] != 0;
      const bool vF = role > 0 && r < RB && r < nreal && col <= a.mc && act_r;
      fl |= vF ? 1u << (24 + u) : 0u;
      const size_t idx = vF ? (col < a.mc ? a.off_E() + size_t(t) * a.mc + col : a.off_g() + t) : a.off_g();
      rw.fraw[u] = Rsel[idx];
      if (BOTH) rw.fraw1[u] = Rsel[alt + idx];
    }
    rw.flags = fl;
  };
  // (every load of the block is requested before the first store: the damping of a diagonal entry is filed -- a.dadd -- as it
  //  is formed, and as far as the compiler knows that store may alias a.R)
  auto fetch_finish = [&](int i, auto& rw, auto& pr, auto both_tag, const auto& m) {
    typedef typename std::remove_reference<decltype(m)>::type M;
    constexpr int NU = M::NU, NF = M::NF, NUD = M::NUD;
    constexpr bool BOTH = decltype(both_tag)::value;
    const int blk = blk0 + i;
    const int nreal = a.n_s() - RB * blk;
    const unsigned fl = rw.flags;
    if (BOTH) {
      const bool second = r_cur_v != 0;
#pragma unroll
      for (int u = 0; u < NUD; ++u) rw.vraw[u] = second ? rw.vraw1[u] : rw.vraw[u];
#pragma unroll
      for (int u = 0; u < NU; ++u) { rw.graw[u] = second ? rw.graw1[u] : rw.graw[u]; rw.garaw[u] = second ? rw.garaw1[u] : rw.garaw[u]; }
#pragma unroll
      for (int u = 0; u < NF; ++u) rw.fraw[u] = second ? rw.fraw1[u] : rw.fraw[u];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) { pr.bt[u] = (fl >> (8 + u)) & 1 ? rw.graw[u] : 0.0; pr.at[u] = (fl >> (16 + u)) & 1 ? rw.garaw[u] : 0.0; }
#pragma unroll
    for (int u = 0; u < NUD; ++u) {
      const int r = m.rcD[u] >> 8, c = m.rcD[u] & 255;
      const bool vD = (fl >> u) & 1;
      double v = vD ? rw.vraw[u] : 0.0;
      {
        // LM damping of a diagonal entry (FromR::damping), for every entry and selected afterwards (no branch around
        // loads). 1 / (radius s^2) as a product of 1 / radius (once per thread) and the filed 1 / s^2; only a solve's
        // first linear solve, which forms the scale itself, divides.
        const bool dg = r == c;
        const int t = RB * blk + r;
        const bool real_row = r < RB && r < nreal;
        double d;
        const double inv_radius = 1.0 / radius;       // (once per thread: the same value in every entry)
        if (fr.first_scale == 0) d = fmin(fmax(v * rw.svv[u] * rw.svv[u], o.min_lm_diagonal), o.max_lm_diagonal) * (inv_radius * rw.q2v[u]);
        else d = fr.damping(v, (dg && real_row) ? t : 0);
        if (dg) {
          if (vD) { v += d; if (role == 0) a.dadd[t] = d; }
          else { v = 1.0; if (role == 0 && real_row) a.dadd[t] = 0.0; }
        }
      }
      pr.d[u] = v;
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) pr.f[u] = (fl >> (24 + u)) & 1 ? rw.fraw[u] : 0.0;
  };
  auto fetch = [&](int i, auto& pr, auto both_tag, const auto& m) {
    typedef typename std::remove_reference<decltype(m)>::type M;
    constexpr int NL = M::NL, NU = M::NU, NF = M::NF;
    const int lt = m.lt;
    const int blk = blk0 + i, mask = pend_mask;
    const bool has_next = (i + 1 < q) || right >= 0;
    const bool has_a = (i == 0) && left >= 0;
    const size_t lc = size_t(left > 0 ? left : 0);
    if (FROM_R) {
      BcrRaw<M::NUD, NU, NF> rw;
      fetch_request(i, rw, both_tag, m);
      fetch_finish(i, rw, pr, both_tag, m);
      return;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = min(max(lt, 0) + NL * u, BB - 1);
      const double v0 = b.D[size_t(blk) * BB + e];
      const double v1 = pendD_r[(size_t(blk) * 2 + 0) * BB + e];
      const double v2 = pendD_r[(size_t(blk) * 2 + 1) * BB + e];
      pr.d[u] = (v0 + ((mask & 1) ? v1 : 0.0)) + ((mask & 2) ? v2 : 0.0);
      const double g = Gr[size_t(blk) * BB + e];
      pr.bt[u] = has_next ? g : 0.0;
      double ga = 0.0;
      if (has_a) ga = Gr[lc * BB + e];
      pr.at[u] = ga;
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = min(max(lt, 0) + NL * u, BP * kBcrFS - 1);
      const int r = e >> 4, j = e & 15;
      const int col = role > 0 ? f0 + j : 0;
      const size_t o2 = size_t(r) * m1p + col;
      const double v0 = b.F[size_t(blk) * fblk + o2];
      const double v1 = pendF_r[(size_t(blk) * 2 + 0) * fblk + o2];
      const double v2 = pendF_r[(size_t(blk) * 2 + 1) * fblk + o2];
      pr.f[u] = role > 0 ? (v0 + ((mask & 1) ? v1 : 0.0)) + ((mask & 2) ? v2 : 0.0) : 0.0;
    }
  };
  auto commit = [&](int p, const auto& pr, const auto& m) {
    typedef typename std::remove_reference<decltype(m)>::type M;
    constexpr int NL = M::NL, NU = M::NU, NF = M::NF;
    const int lt = m.lt;
    double* Dp = Daug + p * 64 * DLD;
    double* Xp = Xb + p * BP * XLD;
    if (FROM_R) {
#pragma unroll
      for (int u = 0; u < M::NUD; ++u) {      // D: the upper triangle as fetched, mirrored (the block is symmetric)
        if (lt + NL * u < kBcrTri) {
          const int r = m.rcD[u] >> 8, c = m.rcD[u] & 255;
          Dp[r * DLD + c] = pr.d[u];
          if (r != c) Dp[c * DLD + r] = pr.d[u];
          if (!ELIM) { Dp[(BP + r) * DLD + c] = r == c ? 1.0 : 0.0; if (r != c) Dp[(BP + c) * DLD + r] = 0.0; }
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int e = lt + NL * u;
        if (e < BB) {
          const int c = e >> 5, r = e & 31;     // (level 0 deals B / A by columns: see BcrLoadMap)
          Xp[c * XLD + CB + r] = pr.bt[u];     // B[row of this block][next's dim] = G[next's dim][row]
          Xp[r * XLD + CA + c] = pr.at[u];     // A[row of this block][left separator's dim] = G_left as stored
        }
      }
    } else {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = lt + NL * u;
      if (e < BB) {
        const int r = e >> 5, c = e & 31;
        Dp[r * DLD + c] = pr.d[u];
        if (!ELIM) Dp[(BP + r) * DLD + c] = r == c ? 1.0 : 0.0;      // (the block elimination forms the identity rows in registers; its L⁻ᵀ in these rows is still being filed when the next commit comes)
        Xp[c * XLD + CB + r] = pr.bt[u];     // B[row of this block][next's dim] = G[next's dim][row]
        Xp[r * XLD + CA + c] = pr.at[u];     // A[row of this block][left separator's dim] = G_left as stored
      }
    }
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = lt + NL * u;
      if (e < BP * kBcrFS) Xp[(e >> 4) * XLD + CF + (e & 15)] = pr.f[u];
    }
  };
  const bool dbg = CAL_DEV_TIMING(a.debug && a.debug < 4 && bid < 2 && lane == 0 && (wave == 0 || wave == 5));
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = dbg ? __builtin_readcyclecounter() : 0;
  const long long t_setup = CAL_DEV_TIMING(a.debug >= 4) ? __builtin_readcyclecounter() - t_kernel : 0;
  long long t_first = 0, t_loop = 0, t_top[4] = {0, 0, 0, 0}, t_elim[4] = {0, 0, 0, 0}, t_bara[4] = {0, 0, 0, 0};
#define LTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  double pmin = 1.0;
  // U_aa (role 0: waves 0..3, tile (wave >> 1, wave & 1)) or U_aF (border roles: waves 2, 3, row tile wave - 2)
  f64x4 acc_a = {0.0, 0.0, 0.0, 0.0};
  if constexpr (ROLL) {
    // ================================================================================================================
    // The rolling chief (round 6). A chain step used to be: elimination (chief 6.8k clocks, followers 0.7k behind) | barrier |
    // filing + the Schur updates of the next block out of LDS (3.4k) | barrier -- 12k clocks, of which the dependent chain
    // is the chief's 6.8k --, behind a head of 12-15k clocks (state -> loads -> staging in LDS -> barrier -> first pivot). Here
    // nothing of the chain waits at a barrier:
    //   * waves 0 and 1 take turns as the chief. While one of them factors block k, the other follows it with the rows of
    //     Bᵀ = T(k+1, k)ᵀ (-> Z^B) and accumulates D_{k+1} -= Z^BᵀZ^B, step by step, in the chief's own register layout
    //     (elim_follow_d): when the chief publishes its last pivot, the follower is one step of products away from holding the
    //     next block's diagonal and goes on as ITS chief (elim_chief_reg); the old chief turns follower of the new one;
    //   * tiles come from the reduce buffer R(x) in the layout their wave holds them in: a lane's entries sit at offsets that
    //     depend on the lane alone (the band is uniform in time; g_roll_tab), so a tile is one load per register, the damping
    //     applied on the way. The first chief and the first follower load their own (both reduce buffers: which one holds
    //     R(x) is a word of the state, still on its way) and start one round trip after the kernel does; the later blocks'
    //     inputs are staged as tile IMAGES in LDS by wave 7, a block ahead -- 28 conflict-free reads for the follower;
    //   * wave 2 follows with the rows of Aᵀ (-> Z^A); its input for block k+1 is the fill -Z^B_kᵀZ^A_k, formed in the registers
    //     the elimination keeps it in as soon as block k's Z^B is complete. Wave 3: the identity rows (role 0: L⁻ᵀ) or the rows of
    //     the role's F slice (-> Z^F), its input F_{k+1} - Z^B_kᵀZ^F_k formed the same way. They start a block behind its chief
    //     and catch up (a follower's step is ~400 clocks, the chief's ~850);
    //   * waves 4..7, once block k's followers are through, file what the back-substitution reads, add up what the left
    //     separator collects, and clear the block's channel.
    // Order is kept by single-writer counters in LDS (a wave's LDS instructions execute in order: data first, then the
    // counter; a reader polls the counter, then reads). Z, L⁻ᵀ and the tile images are double-buffered by parity, the channel three deep.
    // Same products in the same order as the barrier form: bit-identical results (profiles/dev/bitwise.py).
    // Unobserved control points (b.all_active == 0) are padding rows / columns, by a five-bit mask per superblock asked for with its tiles.
    // The levels above level 0 (FROM_R = false; single-block chains: the host sends longer ones to the barrier form): the same waves
    // in the same roles, each taking its tiles straight from D / G / F and the pending slots -- no staging, no barrier but the
    // first --, and the chain's results leave as the barrier form's do (write-through where the Schur complement's riders read them).
    // ================================================================================================================
    constexpr int kImg = 28 * 64;                         // a follower's inputs: spine (12 registers) + rows of Bᵀ (16), by lane
    double* const Zr = lds;                               // [2][32·XLD] Z = [Z^A | Z^B | Z^F] by block parity
    double* const Mr = Zr + 2 * BP * XLD;                 // [2][32·DLD] L⁻ᵀ (role 0)
    double* const chb = Mr + 2 * BP * DLD;                // [3][kElimBufDoubles] channels, block k's: k % 3 (a chief never waits for its channel:
                                                          // the one it writes was cleared behind block k - 3, and the wave saw that as block k - 1's follower)
    double* const img = chb + 3 * kElimBufDoubles;        // [2][kImg] tile images
    int* const ctr = reinterpret_cast<int*>(img + 2 * kImg);                // [16] counters
    long long* const tstamp = reinterpret_cast<long long*>(ctr + 16);       // [8 waves][4 blocks][2] (dev timing)
    static_assert(2 * BP * XLD + 2 * BP * DLD + 3 * kElimBufDoubles + 2 * kImg + 8 + 64 + 16 <= 2 * 64 * DLD + 4 * BP * XLD + 80 + 128 + kLevelThreads + kElimBufDoubles,
                  "the rolling chief's LDS layout must fit the level kernels' allocation");
    enum { C_DONE_D0 = 0, C_DONE_D1 = 1, C_DONE_A = 2, C_DONE_F = 3, C_TAKEN_A = 4, C_TAKEN_F = 5, C_TAKEN_D0 = 6, C_TAKEN_D1 = 7, C_FILED = 8, C_STAGED = 12 };
    auto ctr_set = [&](int idx, int v) {
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(ctr + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
    };
    // cond(c): c(i) = counter i (wave-uniform); polls until it holds
    auto ctr_wait = [&](auto cond) {
      for (;;) {
        const int cv = __hip_atomic_load(ctr + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cond([&](int i) { return __builtin_amdgcn_readlane(cv, i); })) break;
        __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
    };
    auto all4_ge = [&](auto& c, int base, int v) { return c(base) >= v && c(base + 1) >= v && c(base + 2) >= v && c(base + 3) >= v; };
    auto done_d = [&](auto& c, int j) { return c((j & 1) ? C_DONE_D0 : C_DONE_D1) >= j + 1; };      // block j's Z^B complete (its follower: wave 1 for even j)
    const bool tdbg = CAL_DEV_TIMING(a.debug >= 4);
    auto stamp = [&](int k, int which) { if (tdbg && lane == 0 && k < 4) tstamp[(wave * 4 + k) * 2 + which] = __builtin_readcyclecounter() - t_kernel; };
    long long* const hs = tstamp + 64;      // [2 waves][8] head anatomy of the two chiefs (dev timing)
    auto hstamp = [&](int i) { if (tdbg && lane == 0 && wave < 2) hs[wave * 8 + i] = __builtin_readcyclecounter() - t_kernel; };
    hstamp(0);
    // The launch's ONLY barrier, at its top: channels and counters are clear behind it (all waves arrive within a few hundred
    // clocks of the kernel's start; nobody has waited for anything yet). The lanes' offset tables are requested in front of it.
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    const bool has_left = left >= 0;
    const int n_s = a.n_s(), kk = a.k;
    const bool tile_wave = wave < 2 || (FROM_R && wave == 7);         // the waves that load spines and rows of Bᵀ
    const unsigned tmask = FROM_R ? tv[7].y : 0u;
    auto leave_terminated = [&]() { if (pub) fanin_arrive(fan_word); };      // (every wave of the workgroup takes this exit or none does)
    for (int e = tid; e < 3 * kElimBufDoubles; e += kLevelThreads) reinterpret_cast<unsigned long long*>(chb)[e] = kElimSentinel;
    if (tid < 16) ctr[tid] = 0;
    if (tdbg && tid < 64) tstamp[tid] = 0;
    hstamp(1);
    lds_barrier();
    hstamp(2);
    if (CAL_DEV_TIMING(a.debug == 5)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); hstamp(7); }      // (when have the state and the table arrived?)
    const size_t alt = a.r_stride;                        // the same entry of the other reduce buffer
    const double* const bandR = R_buf0 + a.off_B();       // (+ alt: buffer 1) superblock I's storage starts at I·strideB
    // base + zext(byte offset): the scalar-base + 32-bit-vector-offset form of a global load -- one instruction per request
    auto ldo = [](const double* base, unsigned byte_off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off); };
    // operand of the 16x16x4 products for sixteen columns of a Z buffer: lane (l16, lk) holds Z[lk + 4u][col0 + l16]
    auto zops = [&](const double* Zb, int col0, double (&v)[8]) {
      const double* pp = Zb + lk * XLD + col0 + l16;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pp[4 * u * XLD];
    };
    // One column half (sixteen dimensions of the left separator) of the fill T(right, left) = -Z^BᵀZ^A of the chain's last block -- the
    // left separator's coupling to its next survivor: 16 products. Wave 2 takes half 0, the wave that was the last block's chief (idle
    // by then) half 1: on wave 2 alone the 32 products were the tail of the workgroup that ends the launch.
    auto fill_half = [&](int jt) {
      ctr_wait([&](auto c) { return done_d(c, q - 1) && c(C_DONE_A) >= q; });
      const double* Zp = Zr + ((q - 1) & 1) * BP * XLD;
      double zb0[8], zb1[8], za[8];
      zops(Zp, CB, zb0); zops(Zp, CB + 16, zb1); zops(Zp, CA + 16 * jt, za);
      f64x4 g0 = zero4, g1 = zero4;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        g0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-zb0[u], za[u], g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-zb1[u], za[u], g1, 0, 0, 0);
      }
      double* dst = Gw + size_t(left) * BB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dst[(lk + 4 * r) * BP + 16 * jt + l16] = g0[r];
        dst[(16 + lk + 4 * r) * BP + 16 * jt + l16] = g1[r];
      }
    };
    const bool want_fill = right >= 0 && has_left && role == 0;
    // Unobserved control points (b.all_active == 0; e.g. at the ends of a trajectory longer than its data): their rows and columns
    // are padding -- 1 on the diagonal, 0 elsewhere, like FromR::trow. A superblock's five flags are asked for with its tiles (lane
    // j < 5: control point 5 I + j) and become a wave-uniform mask when the tiles are taken; row r of a superblock belongs to control
    // point r / 6 (rows 30, 31: bit 5, never set).
    const bool all_act = b.all_active != 0;
    auto req_act = [&](int I) { return all_act ? 1 : int(a.cp_active[min(max(kBcrCps * I + (lane & 7), 0), a.n_cp - 1)]); };
    auto act_mask = [&](int flag) { return all_act ? 31u : (unsigned(__builtin_amdgcn_ballot_w64(flag != 0 && (lane & 7) < kBcrCps)) & 31u); };
    auto act = [](unsigned am, int row) { return ((am >> (row / 6)) & 1u) != 0; };
    if (tile_wave) {
      if constexpr (!FROM_R) {
        // ---- an upper level's two chief-side waves: wave 0 factors the (single) block, wave 1 follows with the rows of Bᵀ and
        //      leaves -Z^BᵀZ^B in the right separator's pending slot (elim_follow_d against a zero diagonal) ----
        const int blk = blk0, mask = pend_mask;
        const ElimChannel chk = elim_channel(chb);
        auto summed = [&](const double* base, const double* p0, const double* p1, int o) {
          const double v0 = base[o], v1 = p0[o], v2 = p1[o];
          return (v0 + ((mask & 1) ? v1 : 0.0)) + ((mask & 2) ? v2 : 0.0);
        };
        if (wave == 0) {
          const double* const Db = b.D + size_t(blk) * BB;
          const double* const P0 = pendD_r + (size_t(blk) * 2 + 0) * BB;
          const double* const P1 = pendD_r + (size_t(blk) * 2 + 1) * BB;
          f64x4 t00, t01, t11;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = lk + 4 * r, hi = max(l16, c), lo = min(l16, c);
            t00[r] = -summed(Db, P0, P1, hi * BP + lo);
            t01[r] = -summed(Db, P0, P1, (16 + l16) * BP + c);
            t11[r] = -summed(Db, P0, P1, (16 + hi) * BP + 16 + lo);
          }
          if (uniform(terminated_v)) { leave_terminated(); return; }
          stamp(0, 0);
          elim_chief_reg<0>(t00, t01, t11, nullptr, 0, chk, lane);
          stamp(0, 1);
        } else {
          const bool has_next = right >= 0;
          const double* const Gb = Gr + size_t(blk) * BB;      // G[next's dim][this block's dim]
          f64x4 x0[2], x1[2], n00 = zero4, n01 = zero4, n11 = zero4;
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double g0 = Gb[(16 * qt + l16) * BP + lk + 4 * r], g1 = Gb[(16 * qt + l16) * BP + 16 + lk + 4 * r];
              x0[qt][r] = -(has_next ? g0 : 0.0); x1[qt][r] = -(has_next ? g1 : 0.0);
            }
          }
          if (uniform(terminated_v)) { leave_terminated(); return; }
          stamp(0, 0);
          elim_follow_d(x0, x1, Zr + CB, Zr + CB + 16, 1, XLD, chk, lane, n00, n01, n11);
          ctr_set(C_DONE_D1, 1);
          stamp(0, 1);
          if (right >= 0 && role == 0) {
            double* dst = pendD_w + (size_t(right) * 2 + 0) * BB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int c = lk + 4 * r;
              put(dst + c * BP + l16, -n00[r]);
              put(dst + (16 + l16) * BP + c, -n01[r]);
              put(dst + (16 + c) * BP + 16 + l16, -n11[r]);
            }
          }
        }
      } else {
      // this lane's entries of a spine (tiles (0,0), (0,1), (1,1): elim_load_spine) and of the rows of Bᵀ (two row tiles of
      // sixteen dimensions of the next block against this block's 32 columns: elim_load_rows) in the band's storage
      unsigned oS[12], oB[16];
#pragma unroll
      for (int i = 0; i < 3; ++i) { oS[4 * i] = tv[i].x; oS[4 * i + 1] = tv[i].y; oS[4 * i + 2] = tv[i].z; oS[4 * i + 3] = tv[i].w; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { oB[4 * i] = tv[3 + i].x; oB[4 * i + 1] = tv[3 + i].y; oB[4 * i + 2] = tv[3 + i].z; oB[4 * i + 3] = tv[3 + i].w; }
      const unsigned okS = tv[7].x & 0xffffu, okB = tv[7].x >> 16;
      const bool diag_lane = l16 >= lk && ((l16 - lk) & 3) == 0;      // register (l16 - lk) / 4 of tiles (0,0) and (1,1) is a diagonal entry
      struct Inputs { double sp[12], sc[4], bt[16]; int fa, fn; };      // fa / fn: the activity flags of the rows' superblock J and of J + 1 (the spine's)
      // requests: spine of superblock I, the Jacobi scales of its diagonal entries, rows of Bᵀ of superblock J
      auto req_scale = [&](int I, double (&sc)[4]) {
        const int nreal = n_s - RB * I;
        const unsigned t0 = (diag_lane && l16 < nreal) ? RB * I + l16 : 0, t1 = (diag_lane && 16 + l16 < RB && 16 + l16 < nreal) ? RB * I + 16 + l16 : 0;
        sc[0] = ldo(a.scale, 8u * t0); sc[1] = ldo(a.scale + a.NT(), 8u * t0); sc[2] = ldo(a.scale, 8u * t1); sc[3] = ldo(a.scale + a.NT(), 8u * t1);
      };
      auto req_spine = [&](int I, double (&sp)[12], const double* base) {
        const double* p = base + size_t(I) * strideB;
#pragma unroll
        for (int e = 0; e < 12; ++e) sp[e] = ldo(p, oS[e]);
      };
      auto req_b = [&](int J, double (&bt)[16], const double* base) {
        const double* p = base + size_t(J) * strideB;
#pragma unroll
        for (int e = 0; e < 16; ++e) bt[e] = ldo(p, oB[e]);
      };
      // what arrived -> tiles. Spine: structure, the trajectory's end, LM damping of the diagonal (FromR::diag_block); role 0 files dadd.
      const double inv_radius = 1.0 / radius;
      // (ACT: with the activity tests compiled in; where every control point is observed -- a wave-uniform branch at the call -- the tests
      //  are not there at all: a hundred instructions of cold code less in front of the first pivot)
      auto take_spine_t = [&](auto act_tag, int I, const double (&sp)[12], const double (&sc)[4], unsigned am, f64x4& t00, f64x4& t01, f64x4& t11) {
        constexpr bool ACT = decltype(act_tag)::value;
        auto act = [](unsigned m_, int row) { return !ACT || ((m_ >> (row / 6)) & 1u) != 0; };
        const int nreal = n_s - RB * I;
        double e00[4], e01[4], e11[4];
        const bool a0 = act(am, l16), a1 = act(am, 16 + l16);      // this lane's rows l16 and 16 + l16
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = lk + 4 * r, hi = max(l16, c);
          const bool v00 = ((okS >> r) & 1) && hi < nreal && a0 && act(am, c), v01 = ((okS >> (4 + r)) & 1) && 16 + l16 < nreal && a1 && act(am, c),
                     v11 = ((okS >> (8 + r)) & 1) && 16 + hi < nreal && a1 && act(am, 16 + c);
          e00[r] = v00 ? sp[r] : 0.0; e01[r] = v01 ? sp[4 + r] : 0.0; e11[r] = v11 ? sp[8 + r] : 0.0;
        }
        // the lane's diagonal entries (rows l16 and 16 + l16: register (l16 - lk) / 4 of tiles (0,0) and (1,1) in the lanes that
        // have one), ONCE per lane and without a branch on the register: damped where the row is real, 1 on padding rows
        const int dr = (l16 - lk) >> 2;
        double dg[2];
        dg[0] = dr == 1 ? e00[1] : (dr == 2 ? e00[2] : (dr == 3 ? e00[3] : e00[0]));
        dg[1] = dr == 1 ? e11[1] : (dr == 2 ? e11[2] : (dr == 3 ? e11[3] : e11[0]));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = 16 * h + l16, t = RB * I + row;
          const bool in_traj = row < RB && row < nreal, real_row = in_traj && (h == 0 ? a0 : a1);
          double d;
          if (fr.first_scale == 0) d = fmin(fmax(dg[h] * sc[2 * h] * sc[2 * h], o.min_lm_diagonal), o.max_lm_diagonal) * (inv_radius * sc[2 * h + 1]);
          else d = fr.damping(dg[h], real_row ? t : 0);
          if (diag_lane && in_traj && role == 0) a.dadd[t] = real_row ? d : 0.0;
          dg[h] = real_row ? dg[h] + d : 1.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool is_d = diag_lane && dr == r;
          t00[r] = -(is_d ? dg[0] : e00[r]); t01[r] = -e01[r]; t11[r] = -(is_d ? dg[1] : e11[r]);
        }
      };
      auto take_spine = [&](int I, const double (&sp)[12], const double (&sc)[4], unsigned am, f64x4& t00, f64x4& t01, f64x4& t11) {
        if (all_act) take_spine_t(std::false_type(), I, sp, sc, am, t00, t01, t11); else take_spine_t(std::true_type(), I, sp, sc, am, t00, t01, t11);
      };
      auto take_b_t = [&](auto act_tag, int k, const double (&bt)[16], unsigned am_this, unsigned am_next, f64x4 (&x0)[2], f64x4 (&x1)[2]) {
        constexpr bool ACT = decltype(act_tag)::value;
        auto act = [](unsigned m_, int row) { return !ACT || ((m_ >> (row / 6)) & 1u) != 0; };
        const bool has_next = (k + 1 < q) || right >= 0;
        const int nreal_n = n_s - RB * (blk0 + k + 1);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const bool row_ok = has_next && 16 * qt + l16 < nreal_n && act(am_next, 16 * qt + l16);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            x0[qt][r] = -((((okB >> ((qt * 2 + 0) * 4 + r)) & 1) && row_ok && act(am_this, lk + 4 * r)) ? bt[(qt * 2 + 0) * 4 + r] : 0.0);
            x1[qt][r] = -((((okB >> ((qt * 2 + 1) * 4 + r)) & 1) && row_ok && act(am_this, 16 + lk + 4 * r)) ? bt[(qt * 2 + 1) * 4 + r] : 0.0);
          }
        }
      };
      auto take_b = [&](int k, const double (&bt)[16], unsigned am_this, unsigned am_next, f64x4 (&x0)[2], f64x4 (&x1)[2]) {
        if (all_act) take_b_t(std::false_type(), k, bt, am_this, am_next, x0, x1); else take_b_t(std::true_type(), k, bt, am_this, am_next, x0, x1);
      };
      if (wave < 2) {
        // ---- the two chiefs ----
        const int par = wave;
        f64x4 t00 = zero4, t01 = zero4, t11 = zero4;
        f64x4 x0[2], x1[2], n00 = zero4, n01 = zero4, n11 = zero4;      // wave 1: what it follows block 0 with
        // ---- head: the first chief's spine / the first follower's rows of Bᵀ and the second block's spine, from both buffers ----
        {
          Inputs in, alt_in;
          if (par == 0) {
            req_spine(blk0, in.sp, bandR); req_spine(blk0, alt_in.sp, bandR + alt); req_scale(blk0, in.sc);
            in.fn = req_act(blk0); in.fa = in.fn;
          } else {
            req_b(blk0, in.bt, bandR); req_b(blk0, alt_in.bt, bandR + alt);
            if (q > 1) { req_spine(blk0 + 1, in.sp, bandR); req_spine(blk0 + 1, alt_in.sp, bandR + alt); req_scale(blk0 + 1, in.sc); }
            in.fa = req_act(blk0); in.fn = req_act(blk0 + 1);
          }
          hstamp(3);
          if (uniform(terminated_v)) { leave_terminated(); return; }
          hstamp(4);
          const bool second = uniform(r_cur_v) != 0;
#pragma unroll
          for (int e = 0; e < 12; ++e) in.sp[e] = second ? alt_in.sp[e] : in.sp[e];
          if (tdbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); hstamp(5); }
          const unsigned ma = act_mask(in.fa), mn = act_mask(in.fn);
          if (par == 0) take_spine(blk0, in.sp, in.sc, mn, t00, t01, t11);
          else {
#pragma unroll
            for (int e = 0; e < 16; ++e) in.bt[e] = second ? alt_in.bt[e] : in.bt[e];
            take_b(0, in.bt, ma, mn, x0, x1);
            if (q > 1) take_spine(blk0 + 1, in.sp, in.sc, mn, n00, n01, n11);
          }
          hstamp(6);
        }
        for (int k = 0; k < q; ++k) {
          const ElimChannel chk = elim_channel(chb + (k % 3) * kElimBufDoubles);
          if ((k & 1) == par) {
            stamp(k, 0);
            elim_chief_reg<0>(t00, t01, t11, nullptr, 0, chk, lane);
            stamp(k, 1);
          } else {
            if (k > 0) {
              // the tile images wave 7 staged for this block: spine of block k + 1 (zeros behind the chain's last block), rows of Bᵀ
              ctr_wait([&](auto c) { return c(C_STAGED) >= k + 1 && (k < 2 || (all4_ge(c, C_FILED, k - 1) && c(C_TAKEN_A) >= k && (role == 0 || c(C_TAKEN_F) >= k))); });
              const double* im = img + (k & 1) * kImg + lane;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                n00[r] = im[(0 + r) * 64]; n01[r] = im[(4 + r) * 64]; n11[r] = im[(8 + r) * 64];
                x0[0][r] = im[(12 + r) * 64]; x1[0][r] = im[(16 + r) * 64]; x0[1][r] = im[(20 + r) * 64]; x1[1][r] = im[(24 + r) * 64];
              }
              ctr_set(par == 0 ? C_TAKEN_D0 : C_TAKEN_D1, k + 1);
            }
            stamp(k, 0);
            double* const Zk = Zr + (k & 1) * BP * XLD;
            elim_follow_d(x0, x1, Zk + CB, Zk + CB + 16, 1, XLD, chk, lane, n00, n01, n11);
            ctr_set(par == 0 ? C_DONE_D0 : C_DONE_D1, k + 1);
            stamp(k, 1);
            if (k + 1 < q) { t00 = n00; t01 = n01; t11 = n11; }
            else if (right >= 0 && role == 0) {
              // pending D of the right separator, from its left (side 0): -Z^BᵀZ^B; the tiles hold +Z^BᵀZ^B. Tiles (0,0), (1,0), (1,1)
              // (nobody reads the upper right one); the diagonal tiles are symmetric: written along the rows.
              double* dst = pendD_w + (size_t(right) * 2 + 0) * BB;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int c = lk + 4 * r;
                dst[c * BP + l16] = -n00[r];
                dst[(16 + l16) * BP + c] = -n01[r];
                dst[(16 + c) * BP + 16 + l16] = -n11[r];
              }
            }
          }
        }
        if (want_fill && par == ((q - 1) & 1)) fill_half(1);      // (the last block's chief: see fill_half)
      } else {
        // ---- wave 7: stages the followers' inputs of blocks 1.. as tile images, a block ahead; files with waves 4..6 ----
        if (uniform(terminated_v)) { leave_terminated(); return; }
        a.R = uniform(r_cur_v) ? R_buf1 : R_buf0;
        const double* const bandC = a.R + a.off_B();      // R(x)'s band
        const int lt2 = tid - 256;
        for (int k = 0; k < q; ++k) {
          const int j = k + 1;
          if (j < q) {
            stamp(k, 0);
            Inputs in;
            req_b(blk0 + j, in.bt, bandC);
            if (j + 1 < q) { req_spine(blk0 + j + 1, in.sp, bandC); req_scale(blk0 + j + 1, in.sc); }
            in.fa = req_act(blk0 + j); in.fn = req_act(blk0 + j + 1);
            f64x4 x0[2], x1[2], n00 = zero4, n01 = zero4, n11 = zero4;
            const unsigned ma = act_mask(in.fa), mn = act_mask(in.fn);
            take_b(j, in.bt, ma, mn, x0, x1);
            if (j + 1 < q) take_spine(blk0 + j + 1, in.sp, in.sc, mn, n00, n01, n11);
            // (the image of block j - 2 has been taken: its follower is this block's)
            if (j >= 3) ctr_wait([&](auto c) { return c((j & 1) ? C_TAKEN_D0 : C_TAKEN_D1) >= j - 1; });
            double* im = img + (j & 1) * kImg + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              im[(0 + r) * 64] = n00[r]; im[(4 + r) * 64] = n01[r]; im[(8 + r) * 64] = n11[r];
              im[(12 + r) * 64] = x0[0][r]; im[(16 + r) * 64] = x1[0][r]; im[(20 + r) * 64] = x0[1][r]; im[(24 + r) * 64] = x1[1][r];
            }
            ctr_set(C_STAGED, j + 1);
            stamp(k, 1);
          }
          ctr_wait([&](auto c) { return done_d(c, k) && c(C_DONE_A) >= k + 1 && c(C_DONE_F) >= k + 1; });
          const int blk = blk0 + k;
          const double* const Zk = Zr + (k & 1) * BP * XLD;
          if (role == 0) {
            const double* const Mk = Mr + (k & 1) * BP * DLD;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int e = lt2 + 256 * u, r = e >> 5, c = e & 31;
              b.M[size_t(blk) * BB + e] = Mk[r * DLD + c];
              b.ZA[size_t(blk) * BB + e] = Zk[r * XLD + CA + c];
              b.ZB[size_t(blk) * BB + e] = Zk[r * XLD + CB + c];
            }
          } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int e = lt2 + 256 * u; b.Y[size_t(blk) * fblk + size_t(e >> 4) * m1p + f0 + (e & 15)] = Zk[(e >> 4) * XLD + CF + (e & 15)]; }
          }
          {
            unsigned long long* cp = reinterpret_cast<unsigned long long*>(chb + (k % 3) * kElimBufDoubles);
            for (int e = lt2; e < kElimBufDoubles; e += 256) cp[e] = kElimSentinel;
          }
          ctr_set(C_FILED + 3, k + 1);
        }
      }
      }
    } else if (wave == 2) {
      // ---- the rows of Aᵀ: Z^A. First block: T(block, left separator) from R(x); from the second on the fill -Z^BᵀZ^A ----
      f64x4 pre0[2] = {zero4, zero4}, pre1[2] = {zero4, zero4};
      if constexpr (FROM_R) {
        // entry (this block's row rb, the left separator's column cs): tile qt holds cs = 16 qt + l16, register r rb = 16 h + lk + 4 r
        // (g_roll_tab words 32..47, counted from the left separator's superblock)
        double av[16], av1[16];
        unsigned oA[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { oA[4 * i] = tva[i].x; oA[4 * i + 1] = tva[i].y; oA[4 * i + 2] = tva[i].z; oA[4 * i + 3] = tva[i].w; }
        const double* p = bandR + size_t(max(blk0 - 1, 0)) * strideB;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          av[e] = 0.0; av1[e] = 0.0;
          if (has_left) { av[e] = ldo(p, oA[e]); av1[e] = ldo(p + alt, oA[e]); }
        }
        const int f_this = req_act(blk0), f_left = req_act(blk0 - 1);
        if (uniform(terminated_v)) { leave_terminated(); return; }
        const bool second = uniform(r_cur_v) != 0;
        const int nreal = n_s - RB * blk0;
        const unsigned m_this = act_mask(f_this), m_left = act_mask(f_left);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e0 = (qt * 2 + 0) * 4 + r, e1 = (qt * 2 + 1) * 4 + r;
            const bool col_ok = has_left && act(m_left, 16 * qt + l16);
            const bool v0 = ((tmask >> e0) & 1) && col_ok && lk + 4 * r < nreal && act(m_this, lk + 4 * r),
                       v1 = ((tmask >> e1) & 1) && col_ok && 16 + lk + 4 * r < nreal && act(m_this, 16 + lk + 4 * r);
            pre0[qt][r] = -(v0 ? (second ? av1[e0] : av[e0]) : 0.0);
            pre1[qt][r] = -(v1 ? (second ? av1[e1] : av[e1]) : 0.0);
          }
        }
      } else {
        // G[left separator][this block's dim][the separator's dim]
        const double* const Ga = Gr + size_t(left > 0 ? left : 0) * BB;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double g0 = Ga[(lk + 4 * r) * BP + 16 * qt + l16], g1 = Ga[(16 + lk + 4 * r) * BP + 16 * qt + l16];
            pre0[qt][r] = -(has_left ? g0 : 0.0); pre1[qt][r] = -(has_left ? g1 : 0.0);
          }
        }
        if (uniform(terminated_v)) { leave_terminated(); return; }
      }
      auto fill_pre = [&](const double* Zp) {      // pre = -(0 - Z^BᵀZ^A), in the elimination's (negated) tile layout
        double zb0[8], zb1[8], za[8];
        zops(Zp, CB, zb0); zops(Zp, CB + 16, zb1);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          zops(Zp, CA + 16 * jt, za);
#pragma unroll
          for (int r = 0; r < 4; ++r) { pre0[jt][r] = -0.0; pre1[jt][r] = -0.0; }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pre0[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb0[u], za[u], pre0[jt], 0, 0, 0);
            pre1[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb1[u], za[u], pre1[jt], 0, 0, 0);
          }
        }
      };
      for (int k = 0; k < q; ++k) {
        const ElimChannel chk = elim_channel(chb + (k % 3) * kElimBufDoubles);
        double* const Zk = Zr + (k & 1) * BP * XLD;
        if (k > 0) {
          ctr_wait([&](auto c) { return done_d(c, k - 1); });
          stamp(k, 0);
          if (has_left) fill_pre(Zr + ((k - 1) & 1) * BP * XLD);
        } else stamp(k, 0);
        ctr_set(C_TAKEN_A, k + 1);
        if (k >= 2) ctr_wait([&](auto c) { return all4_ge(c, C_FILED, k - 1); });
        const ElimTile t[2] = {{Zk, 0, 0, Zk + CA, 1, XLD, 0, nullptr}, {Zk, 0, 0, Zk + CA + 16, 1, XLD, 0, nullptr}};
        elim_follow<2>(t, chk, lane, true, pre0, pre1);
        ctr_set(C_DONE_A, k + 1);
        stamp(k, 1);
      }
      if (want_fill) { fill_half(0); if (!FROM_R) fill_half(1); }
    } else if (wave == 3) {
      if (role == 0) {
        // ---- the identity rows: L⁻ᵀ ----
        if (uniform(terminated_v)) { leave_terminated(); return; }
        for (int k = 0; k < q; ++k) {
          const ElimChannel chk = elim_channel(chb + (k % 3) * kElimBufDoubles);
          double* const Mk = Mr + (k & 1) * BP * DLD;
          if (k >= 2) ctr_wait([&](auto c) { return all4_ge(c, C_FILED, k - 1); });
          stamp(k, 0);
          const ElimTile t[2] = {{Mk, 0, 0, Mk, DLD, 1, 1, nullptr}, {Mk, 0, 0, Mk + 16 * DLD, DLD, 1, 2, nullptr}};
          elim_follow<2>(t, chk, lane);
          ctr_set(C_DONE_F, k + 1);
          stamp(k, 1);
        }
      } else {
        // ---- the rows of the role's F slice: Z^F; input F_k - Z^BᵀZ^F of the block before; F_{k+1} is requested a block ahead ----
        f64x4 pre0[1], pre1[1];
        const int col = f0 + l16;
        auto req_f = [&](int k, double (&fv)[8], const double* Rb) {
          const int nreal = n_s - RB * (blk0 + k);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = 16 * (e >> 2) + lk + 4 * (e & 3), t = RB * (blk0 + k) + r;
            const bool vF = r < RB && r < nreal && col <= a.mc;
            fv[e] = Rb[vF ? (col < a.mc ? a.off_E() + size_t(t) * a.mc + col : a.off_g() + t) : a.off_g()];
          }
        };
        unsigned m_f = 31u;      // activity of the block whose F rows are in fv
        auto f_valid = [&](int k, int e) {
          if (!FROM_R) return true;      // (D / F of a separator hold the padding's values themselves)
          const int r = 16 * (e >> 2) + lk + 4 * (e & 3);
          return r < RB && r < n_s - RB * (blk0 + k) && col <= a.mc && act(m_f, r);
        };
        double fv[8];
        int f_flag = FROM_R ? req_act(blk0) : 1;
        {
          if constexpr (FROM_R) {
            double fv1[8];
            req_f(0, fv, R_buf0); req_f(0, fv1, R_buf1);
            if (uniform(terminated_v)) { leave_terminated(); return; }
            const bool second = uniform(r_cur_v) != 0;
            a.R = second ? R_buf1 : R_buf0;
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = second ? fv1[e] : fv[e];
          } else {
            // F + what the chains on either side left for this separator (the sum the barrier form's loaders take)
            const double* const Fb = b.F + size_t(blk0) * fblk;
            const double* const P0 = pendF_r + (size_t(blk0) * 2 + 0) * fblk;
            const double* const P1 = pendF_r + (size_t(blk0) * 2 + 1) * fblk;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const size_t o2 = size_t(16 * (e >> 2) + lk + 4 * (e & 3)) * m1p + col;
              const double v0 = Fb[o2], v1 = P0[o2], v2 = P1[o2];
              fv[e] = (v0 + ((pend_mask & 1) ? v1 : 0.0)) + ((pend_mask & 2) ? v2 : 0.0);
            }
            if (uniform(terminated_v)) { leave_terminated(); return; }
          }
        }
        for (int k = 0; k < q; ++k) {
          const ElimChannel chk = elim_channel(chb + (k % 3) * kElimBufDoubles);
          double* const Zk = Zr + (k & 1) * BP * XLD;
          if (k > 0) ctr_wait([&](auto c) { return done_d(c, k - 1); });
          stamp(k, 0);
          m_f = act_mask(f_flag);
#pragma unroll
          for (int r = 0; r < 4; ++r) { pre0[0][r] = -(f_valid(k, r) ? fv[r] : 0.0); pre1[0][r] = -(f_valid(k, 4 + r) ? fv[4 + r] : 0.0); }
          if (FROM_R && k + 1 < q) { req_f(k + 1, fv, a.R); f_flag = req_act(blk0 + k + 1); }
          if (k > 0) {
            const double* Zp = Zr + ((k - 1) & 1) * BP * XLD;
            double zb0[8], zb1[8], zf[8];
            zops(Zp, CB, zb0); zops(Zp, CB + 16, zb1); zops(Zp, CF, zf);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              pre0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb0[u], zf[u], pre0[0], 0, 0, 0);
              pre1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb1[u], zf[u], pre1[0], 0, 0, 0);
            }
          }
          ctr_set(C_TAKEN_F, k + 1);
          if (k >= 2) ctr_wait([&](auto c) { return all4_ge(c, C_FILED, k - 1); });
          const ElimTile t[1] = {{Zk, 0, 0, Zk + CF, 1, XLD, 0, nullptr}};
          elim_follow<1>(t, chk, lane, true, pre0, pre1);
          ctr_set(C_DONE_F, k + 1);
          stamp(k, 1);
        }
        if (right >= 0) {
          // pending F of the right separator: -Z^BᵀZ^F of the chain's last block
          ctr_wait([&](auto c) { return done_d(c, q - 1); });
          const double* Zp = Zr + ((q - 1) & 1) * BP * XLD;
          double zb0[8], zb1[8], zf[8];
          zops(Zp, CB, zb0); zops(Zp, CB + 16, zb1); zops(Zp, CF, zf);
          f64x4 s0 = zero4, s1 = zero4;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-zb0[u], zf[u], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-zb1[u], zf[u], s1, 0, 0, 0);
          }
          double* dst = pendF_w + (size_t(right) * 2 + 0) * fblk + f0 + l16;
#pragma unroll
          for (int r = 0; r < 4; ++r) { put(dst + size_t(lk + 4 * r) * m1p, s0[r]); put(dst + size_t(16 + lk + 4 * r) * m1p, s1[r]); }
        }
      }
    } else {
      // ---- waves 4..6: filing, the left separator's sums, the channel ----
      if (uniform(terminated_v)) { leave_terminated(); return; }
      const int w4 = wave - 4, lt2 = tid - 256;
      const bool acc_owner = wave == 5 || wave == 6, acc_owner2 = role == 0 && wave == 5;
      const int acc_p = role == 0 ? CA + (wave == 6 ? 16 : 0) : CA + 16 * (wave - 5);
      const int acc_q = role == 0 ? CA + (wave == 6 ? 16 : 0) : CF;
      f64x4 acc2 = zero4;
      for (int k = 0; k < q; ++k) {
        ctr_wait([&](auto c) { return done_d(c, k) && c(C_DONE_A) >= k + 1 && c(C_DONE_F) >= k + 1; });
        stamp(k, 0);
        const int blk = blk0 + k;
        const double* const Zk = Zr + (k & 1) * BP * XLD;
        if (role == 0) {
          // (an upper level with riders behind it: what only the back-substitution reads is filed BEHIND the fan-in -- the
          //  arrival drains every store of the wave, and these 24 KB are nobody's business in this launch)
          if (!pub) {
            const double* const Mk = Mr + (k & 1) * BP * DLD;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int e = lt2 + 256 * u, r = e >> 5, c = e & 31;
              b.M[size_t(blk) * BB + e] = Mk[r * DLD + c];
              b.ZA[size_t(blk) * BB + e] = Zk[r * XLD + CA + c];
              b.ZB[size_t(blk) * BB + e] = Zk[r * XLD + CB + c];
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < 2; ++u) { const int e = lt2 + 256 * u; put(b.Y + size_t(blk) * fblk + size_t(e >> 4) * m1p + f0 + (e & 15), Zk[(e >> 4) * XLD + CF + (e & 15)]); }
        }
        if (has_left && acc_owner) acc_a = atb_tile<true>(Zk, XLD, acc_p, Zk, XLD, acc_q, 0, BP, acc_a, lane);
        if (has_left && acc_owner2) acc2 = atb_tile<true>(Zk, XLD, CA + 16, Zk, XLD, CA, 0, BP, acc2, lane);
        // (the followers of block k are through with its channel; block k+2's chief and followers wait for this wave's word)
        {
          unsigned long long* cp = reinterpret_cast<unsigned long long*>(chb + (k % 3) * kElimBufDoubles);
          for (int e = lt2; e < kElimBufDoubles; e += 256) cp[e] = kElimSentinel;
        }
        ctr_set(C_FILED + w4, k + 1);
        stamp(k, 1);
      }
      if (has_left) {
        if (role == 0) {
          if (acc_owner) {
            const int it = wave == 6 ? 1 : 0, jt = it;
            double* dst = pendD_w + (size_t(left) * 2 + 1) * BB;
#pragma unroll
            for (int r = 0; r < 4; ++r) put(dst + (16 * it + lk + 4 * r) * BP + 16 * jt + l16, acc_a[r]);
            if (acc_owner2) {
#pragma unroll
              for (int r = 0; r < 4; ++r) put(dst + (16 + lk + 4 * r) * BP + l16, acc2[r]);
            }
          }
        } else if (acc_owner) {
          const int h = wave - 5;
          double* dst = pendF_w + (size_t(left) * 2 + 1) * fblk + f0 + l16;
#pragma unroll
          for (int r = 0; r < 4; ++r) put(dst + size_t(16 * h + lk + 4 * r) * m1p, acc_a[r]);
        }
      }
    }
    if (tdbg) {
      __syncthreads();
      if (tid == 0 && (bid < 2 || bid == 7 || bid == 8)) {
        printf("bcr_level %d (rolling) wg %d role %d q %d lived %lld clocks: set-up %lld | head of wave 0: entry %lld, at the barrier %lld, behind it %lld, requests out %lld, state there %lld, data there %lld, tiles %lld (all loads there %lld) | wave 1: %lld %lld %lld %lld %lld %lld %lld\n", level, bid, role, q, (long long)(__builtin_readcyclecounter() - t_kernel), t_setup,
               hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7], hs[8], hs[9], hs[10], hs[11], hs[12], hs[13], hs[14]);
        for (int w = 0; w < 8; ++w)
          printf("  wg %d wave %d, begin-end by block: %lld-%lld %lld-%lld %lld-%lld %lld-%lld\n", bid, w, tstamp[(w * 4 + 0) * 2], tstamp[(w * 4 + 0) * 2 + 1],
                 tstamp[(w * 4 + 1) * 2], tstamp[(w * 4 + 1) * 2 + 1], tstamp[(w * 4 + 2) * 2], tstamp[(w * 4 + 2) * 2 + 1], tstamp[(w * 4 + 3) * 2], tstamp[(w * 4 + 3) * 2 + 1]);
      }
    }
    if (pub) {
      fanin_arrive(fan_word);
      if (role == 0 && wave >= 4) {      // (single-block chains: q == 1)
        const int lt2 = tid - 256;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = lt2 + 256 * u, r = e >> 5, c = e & 31;
          b.M[size_t(blk0) * BB + e] = Mr[r * DLD + c];
          b.ZA[size_t(blk0) * BB + e] = Zr[r * XLD + CA + c];
          b.ZB[size_t(blk0) * BB + e] = Zr[r * XLD + CB + c];
        }
      }
    }
    return;
  }
  {
    // ELIM: nobody has anything else to do before the first block is in LDS -- all eight waves fetch it, two entries of
    // D / B / A and one of the F slice each (half the instructions per thread of the loaders' share of a later block)
    typedef typename std::conditional<kAllFetchFirst, BcrLoadMap<kLevelThreads, BB / kLevelThreads, 1>, BcrLoadMap<NL, NU, NF>>::type Map0;
    Map0 m0;
    if constexpr (kAllFetchFirst) { m0.lt = tid; fill_map(m0); } else { m0 = lmap; }
    BcrPre<Map0::NU, Map0::NF> pr;
    const bool load0 = kAllFetchFirst || loader;
    if (load0) fetch(0, pr, std::integral_constant<bool, FROM_R>(), m0);
    if (kAllFetchFirst) fill_map(lmap);
    if (FROM_R) a.R = uniform(r_cur_v) ? R_buf1 : R_buf0;        // (use_current_R; the later blocks' requests come behind the state anyway)
    if (uniform(terminated_v)) { if (pub) fanin_arrive(fan_word); return; }
    if (load0) commit(0, pr, m0);
  }
  asm volatile("" :: "v"(code_pf));      // (prefetch_code: waited for with the first block's loads)
  __syncthreads();
  LTICK(0)
  if (CAL_DEV_TIMING(a.debug >= 4)) t_first = __builtin_readcyclecounter() - t_kernel;
  // (the barriers of the step loop order LDS traffic only: __syncthreads() would also drain the global loads of the
  //  next block, which are meant to stay in flight while this one is factored)
  // LA: the left separator's sums live in loader waves 5 and 6 (wave 4 shares its SIMD with the chief, wave 7 with the
  // follower the next block's diagonal waits for): role 0: wave 5 tiles (0,0) and (1,0) [acc_a, acc_a2], wave 6 tile (1,1);
  // border roles: row tile 0 on wave 5, 1 on wave 6
  const bool la_acc_owner = LA && (wave == 5 || wave == 6), la_acc_owner2 = LA && role == 0 && wave == 5;
  const int la_acc_p = role == 0 ? CA + (wave == 6 ? 16 : 0) : CA + 16 * (wave - 5);
  const int la_acc_q = role == 0 ? CA + (wave == 6 ? 16 : 0) : CF;
  f64x4 acc_a2 = {0.0, 0.0, 0.0, 0.0};
  constexpr bool kEarly = FROM_R && ELIM && !LA && BCR_EARLY_REQUESTS;
  BcrRaw<BcrLoadMap<NL, NU, NF>::NUD, NU, NF> rw_early;
  Pre pr;
  f64x4 pre0[2], pre1[2];       // LA: a follower's own input tiles of the next block, updated (wave 2: the two A tiles; wave 1 of a border role: the F tile)
  if (LA && q > 1 && loader) fetch(1, pr, std::false_type(), lmap);
  const int lane_outer = lane;
  for (int i = 0; i < q; ++i) {
    // (the lane number made opaque once per step: every LDS / global address of a step is lane arithmetic, and kept across
    //  the loop as loop invariants -- some hundred of them over all roles -- they were what the registers ran out on; a few
    //  integer instructions per step form them again)
    int lane_step = lane_outer;
    if (LA) asm volatile("" : "+v"(lane_step));
    const int lane = lane_step, l16 = lane & 15, lk = lane >> 4, tid = 64 * wave + lane;
    const int p = i & 1;
    double* const Zb = Zb_base + (LA ? p * BP * XLD : 0);      // LA: Z of step i is still read while the followers of step i + 1 write theirs
    double* Dp = Daug + p * 64 * DLD;
    double* Xp = Xb + p * BP * XLD;
    double* Dn = Daug + (p ^ 1) * 64 * DLD;
    double* Xn = Xb + (p ^ 1) * BP * XLD;
    const bool last = i + 1 == q;
    const int blk = blk0 + i;
    const long long t_step = CAL_DEV_TIMING(a.debug != 0) ? __builtin_readcyclecounter() : 0;
    if (CAL_DEV_TIMING(a.debug >= 4)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k == i) t_top[k] = t_step - t_kernel;
    }
    if (CAL_DEV_TIMING(a.debug >= 4 && !LA && i == 1 && lane == 0 && wave >= 4)) bcast[16 + wave] = double(__builtin_readcyclecounter() - t_kernel);
    if (!LA && !last && loader) {
      // in flight while the block is factored (LA: requested behind the step before). EARLY (level 0, block elimination): the
      // requests of block i + 1 went out in front of the LAST step's Schur phase -- only the selection, the damping and the
      // commit are left, at once: the loaders' request -> commit time (address arithmetic 3-4k clocks on SIMDs they share with
      // the chief and the followers, then the round trip) was the tail of every step, 1.5-2k clocks behind the followers.
      if (kEarly && i > 0) fetch_finish(i + 1, rw_early, pr, std::false_type(), lmap);
      else fetch(i + 1, pr, std::false_type(), lmap);
    }
    if (CAL_DEV_TIMING(a.debug >= 4 && !LA && i == 1 && lane == 0 && wave >= 4)) bcast[24 + wave] = double(__builtin_readcyclecounter() - t_kernel);
    if (CAL_DEV_TIMING(a.debug == 2 && bid < 1 && lane == 0 && wave >= 4)) printf("level %d step %d wave %d: requests issued at %lld clocks of the step\n", level, i, wave, (long long)(__builtin_readcyclecounter() - t_step));
    if (ELIM) {
      // ---- D = L Lᵀ, Z = L⁻¹X and (role 0) L⁻ᵀ in one pass: wave 0 the spine, waves 1..3 two row tiles each ----
      if (wave == 0) elim_chief<0>(Dp, DLD, ech, lane);
      else if (wave == 1) {
        if (role == 0) {
          const ElimTile t[2] = {{Zb, 0, 0, Dp + BP * DLD, DLD, 1, 1, nullptr}, {Zb, 0, 0, Dp + (BP + 16) * DLD, DLD, 1, 2, nullptr}};
          elim_follow<2>(t, ech, lane);
        } else {
          const ElimTile t[1] = {{Xp + CF, 1, XLD, Zb + CF, 1, XLD, 0, nullptr}};
          elim_follow<1>(t, ech, lane, LA && i > 0, pre0, pre1);
        }
      } else if (wave == 2 || wave == 3) {
        const int c0 = wave == 2 ? CA : CB;
        const ElimTile t[2] = {{Xp + c0, 1, XLD, Zb + c0, 1, XLD, 0, nullptr}, {Xp + c0 + 16, 1, XLD, Zb + c0 + 16, 1, XLD, 0, nullptr}};
        elim_follow<2>(t, ech, lane, LA && wave == 2 && i > 0, pre0, pre1);
      }
      if (CAL_DEV_TIMING(a.debug >= 4 && i == 1 && lane == 0)) bcast[8 + wave] = double(__builtin_readcyclecounter() - t_kernel);
      if (CAL_DEV_TIMING(a.debug == 2 && bid < 1 && lane == 0 && wave < 4)) printf("level %d step %d wave %d: elimination done at %lld clocks of the step\n", level, i, wave, (long long)(__builtin_readcyclecounter() - t_step));
      if (!last && loader) {
        commit(p ^ 1, pr, lmap);
      }
      if (CAL_DEV_TIMING(a.debug >= 4 && i == 1 && lane == 0 && wave >= 4)) bcast[8 + wave] = double(__builtin_readcyclecounter() - t_kernel);
      if (CAL_DEV_TIMING(a.debug == 2 && bid < 1 && lane == 0 && wave >= 4)) printf("level %d step %d wave %d: committed at %lld clocks of the step\n", level, i, wave, (long long)(__builtin_readcyclecounter() - t_step));
      LTICK(3)
    } else {
    // ---- D = L Lᵀ and L⁻ᵀ: two in-wave panels, one tile update between them ----
    if (wave == 0) panel_factor<1, false, false>(Dp, DLD, dinv, bcast, 0, 63, 16, lane, &pmin);
    lds_barrier();
    LTICK(1)
    if (wave < 3) update_tile(Dp, DLD, 63, 1 + wave, 1, 0, 1, lane, dump);
    lds_barrier();
    LTICK(2)
    if (wave == 0) panel_factor<1, false, false>(Dp, DLD, dinv, bcast, 16, 63, 16, lane, &pmin);
    else if (!last && loader) commit(p ^ 1, pr, lmap);
    lds_barrier();
    LTICK(3)
    // ---- Z = L⁻¹ X = MᵀX, M = L⁻ᵀ in rows 32..63 (upper triangular: row tile it needs k < 16(it+1)) ----
    {
      const double* M = Dp + BP * DLD;
      auto zjob = [&](int jt, int it) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = atb_tile<false>(M, DLD, 16 * it, Xp, XLD, 16 * jt, 0, 16 * (it + 1), acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Zb[(16 * it + lk + 4 * r) * XLD + 16 * jt + l16] = acc[r];
      };
      zjob(wave >> 1, wave & 1);
      // the border roles' fifth column tile: its two jobs (4 and 8 MFMAs) go to the SIMDs that carry two 4-MFMA jobs
      // (waves 0/4 and 2/6), not to those with two 8-MFMA jobs: 16 instead of 24 MFMAs on the busiest matrix pipe
      if (role > 0 && (wave == 0 || wave == 2)) zjob(4, wave == 0 ? 1 : 0);
    }
    }
    if (CAL_DEV_TIMING(a.debug >= 4)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k == i) t_elim[k] = __builtin_readcyclecounter() - t_kernel;
    }
    lds_barrier();
    LTICK(4)
    if (CAL_DEV_TIMING(a.debug >= 4)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k == i) t_bara[k] = __builtin_readcyclecounter() - t_kernel;
    }
    if (ELIM && !last) elim_reset(ech, tid, kLevelThreads);      // (the followers are through; the barrier at the end of the step orders it)
    if (kEarly && i + 2 < q && loader) fetch_request(i + 2, rw_early, std::false_type(), lmap);      // (nothing in it waits: see the top of the loop)
    if (LA && !last) {
      // ---- look-ahead: in front of the barrier only next.D -= Z^BᵀZ^B (and the next block's requests); behind it the other
      //      Schur products, read from THIS step's Z buffer (the followers of the next block write the other one) ----
      // operand of the 16x16x4 product for sixteen columns of Z: lane (l16, lk) holds Z[lk + 4u][col0 + l16]
      auto ops = [&](int col0, double (&v)[8]) {
        const double* pp = Zb + lk * XLD + col0 + l16;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[4 * u * XLD];
      };
      if (wave == 0 || wave == 2 || wave == 3) {      // next.D -= Z^BᵀZ^B: tiles (0,0), (1,0), (1,1) -- the upper right one is read by nobody
        const int it = (wave & 3) >> 1, jt = wave & 1;
        const int row0 = 16 * it + lk, col0 = 16 * jt + l16;
        const bool dbgw = CAL_DEV_TIMING(a.debug >= 4 && i == 1 && lane == 0 && wave == 0);
        if (dbgw) bcast[16] = double(__builtin_readcyclecounter() - t_kernel);
        double zp[8], zq[8];
        ops(CB + 16 * it, zp);
        if (it != jt) ops(CB + 16 * jt, zq);
        f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Dn[(row0 + 4 * r) * DLD + col0];
        if (dbgw) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); bcast[17] = double(__builtin_readcyclecounter() - t_kernel); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-zp[u], it != jt ? zq[u] : zp[u], acc, 0, 0, 0);
        if (dbgw) { asm volatile("" :: "v"(acc[0])); bcast[18] = double(__builtin_readcyclecounter() - t_kernel); }
#pragma unroll
        for (int r = 0; r < 4; ++r) Dn[(row0 + 4 * r) * DLD + col0] = acc[r];
      }
      LTICK(5)
      if (CAL_DEV_TIMING(a.debug >= 4 && i == 1 && lane == 0)) bcast[wave] = double(__builtin_readcyclecounter() - t_kernel);
      lds_barrier();
      LTICK(6)
      // the next block but one: requested first thing behind the barrier, in flight while the next block is factored (in
      // front of the barrier its address arithmetic -- 6k clocks at level 0 -- held the chief up)
      if (i + 2 < q && loader) fetch(i + 2, pr, std::false_type(), lmap);
      const bool wF = role > 0 && wave == 1;
      if (wave == 2) {        // next.A -= Z^BᵀZ^A, in the (negated) form the elimination holds its tiles in
        double zb0[8], zb1[8], za[8];
        ops(CB, zb0); ops(CB + 16, zb1);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          ops(CA + 16 * jt, za);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pre0[jt][r] = -Xn[(lk + 4 * r) * XLD + CA + 16 * jt + l16];
            pre1[jt][r] = -Xn[(16 + lk + 4 * r) * XLD + CA + 16 * jt + l16];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            pre0[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb0[u], za[u], pre0[jt], 0, 0, 0);
            pre1[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb1[u], za[u], pre1[jt], 0, 0, 0);
          }
        }
      }
      if (wF) {               // next.F -= Z^BᵀZ^F
        double zb0[8], zb1[8], zf[8];
        ops(CB, zb0); ops(CB + 16, zb1); ops(CF, zf);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pre0[0][r] = -Xn[(lk + 4 * r) * XLD + CF + l16];
          pre1[0][r] = -Xn[(16 + lk + 4 * r) * XLD + CF + l16];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          pre0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb0[u], zf[u], pre0[0], 0, 0, 0);
          pre1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(zb1[u], zf[u], pre1[0], 0, 0, 0);
        }
      }
      // what the back-substitution needs: filed by the loader waves
      if (wave >= 4) {
        const int lt2 = tid - 256;
        if (role == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = lt2 + 256 * u, r = e >> 5, c = e & 31;
            b.M[size_t(blk) * BB + e] = Dp[(BP + r) * DLD + c];
            b.ZA[size_t(blk) * BB + e] = Zb[r * XLD + CA + c];
            b.ZB[size_t(blk) * BB + e] = Zb[r * XLD + CB + c];
          }
        } else {
#pragma unroll
          for (int u = 0; u < 2; ++u) { const int e = lt2 + 256 * u; put(b.Y + size_t(blk) * fblk + size_t(e >> 4) * m1p + f0 + (e & 15), Zb[(e >> 4) * XLD + CF + (e & 15)]); }
        }
      }
      // what the left separator collects over the chain (see la_acc_tile: on loader waves that do not share a SIMD with the chief)
      if (left >= 0 && la_acc_owner) acc_a = atb_tile<true>(Zb, XLD, la_acc_p, Zb, XLD, la_acc_q, 0, BP, acc_a, lane);
      if (left >= 0 && la_acc_owner2) acc_a2 = atb_tile<true>(Zb, XLD, CA + 16, Zb, XLD, CA, 0, BP, acc_a2, lane);
      continue;
    }
    // ---- file what the back-substitution needs ----
    if (role == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + kLevelThreads * u;
        const int r = e >> 5, c = e & 31;
        b.M[size_t(blk) * BB + e] = Dp[(BP + r) * DLD + c];
        b.ZA[size_t(blk) * BB + e] = Zb[r * XLD + CA + c];
        b.ZB[size_t(blk) * BB + e] = Zb[r * XLD + CB + c];
      }
    } else {
      const int r = tid >> 4, j = tid & 15;
      put(b.Y + size_t(blk) * fblk + size_t(r) * m1p + f0 + j, Zb[r * XLD + CF + j]);
    }
    // ---- Schur updates of the next block of the chain (in LDS) or of the right separator (pending slots) ----
    const int it = (wave & 3) >> 1, jt = wave & 1;
    const int row0 = 16 * it + lk, col0 = 16 * jt + l16;
    // (ELIM: nobody reads the upper right 16x16 tile of a diagonal block -- the chief takes the lower triangle and the lower left
    //  tile, the reduced system the lower triangle of the root --, so wave 1's tile of the three symmetric updates is left out)
    const bool sym_skip = ELIM && wave == 1;
    if (!last) {
      if (sym_skip) {
      } else if (wave < 4) {          // next.D -= Z^BᵀZ^B
        f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Dn[(row0 + 4 * r) * DLD + col0];
        acc = atb_tile<true>(Zb, XLD, CB + 16 * it, Zb, XLD, CB + 16 * jt, 0, BP, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Dn[(row0 + 4 * r) * DLD + col0] = acc[r];
      } else {                 // next.A -= Z^BᵀZ^A (fill towards the left separator)
        f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Xn[(row0 + 4 * r) * XLD + CA + col0];
        acc = atb_tile<true>(Zb, XLD, CB + 16 * it, Zb, XLD, CA + 16 * jt, 0, BP, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Xn[(row0 + 4 * r) * XLD + CA + col0] = acc[r];
      }
      if (role > 0 && wave < 2) {   // next.F -= Z^BᵀZ^F
        const int rw = 16 * wave + lk;
        f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Xn[(rw + 4 * r) * XLD + CF + l16];
        acc = atb_tile<true>(Zb, XLD, CB + 16 * wave, Zb, XLD, CF, 0, BP, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Xn[(rw + 4 * r) * XLD + CF + l16] = acc[r];
      }
    } else if (right >= 0) {
      if (role == 0) {
        if (sym_skip) {
        } else if (wave < 4) {        // pending D of the right separator, from its left (side 0)
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          acc = atb_tile<true>(Zb, XLD, CB + 16 * it, Zb, XLD, CB + 16 * jt, 0, BP, acc, lane);
          double* dst = pendD_w + (size_t(right) * 2 + 0) * BB;
#pragma unroll
          for (int r = 0; r < 4; ++r) put(dst + (row0 + 4 * r) * BP + col0, acc[r]);
        } else if (left >= 0) {   // fill T(right, left): the left separator's coupling to its next survivor
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          acc = atb_tile<true>(Zb, XLD, CB + 16 * it, Zb, XLD, CA + 16 * jt, 0, BP, acc, lane);
          double* dst = Gw + size_t(left) * BB;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(row0 + 4 * r) * BP + col0] = acc[r];
        }
      } else if (wave < 2) {   // pending F of the right separator
        const int rw = 16 * wave + lk;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = atb_tile<true>(Zb, XLD, CB + 16 * wave, Zb, XLD, CF, 0, BP, acc, lane);
        double* dst = pendF_w + (size_t(right) * 2 + 0) * fblk + f0 + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) put(dst + size_t(rw + 4 * r) * m1p, acc[r]);
      }
    }
    // ---- what the left separator collects over the chain ----
    if (left >= 0) {
      if (LA) {       // (the owners of the sums are loader waves: see la_acc_owner)
        if (la_acc_owner) acc_a = atb_tile<true>(Zb, XLD, la_acc_p, Zb, XLD, la_acc_q, 0, BP, acc_a, lane);
        if (la_acc_owner2) acc_a2 = atb_tile<true>(Zb, XLD, CA + 16, Zb, XLD, CA, 0, BP, acc_a2, lane);
      } else if (role == 0) {
        if (wave < 4 && !sym_skip) acc_a = atb_tile<true>(Zb, XLD, CA + 16 * it, Zb, XLD, CA + 16 * jt, 0, BP, acc_a, lane);
      } else if (wave == 2 || wave == 3) {
        acc_a = atb_tile<true>(Zb, XLD, CA + 16 * (wave - 2), Zb, XLD, CF, 0, BP, acc_a, lane);
      }
    }
    LTICK(5)
    lds_barrier();
    LTICK(6)
  }
  if (CAL_DEV_TIMING(a.debug >= 4)) t_loop = __builtin_readcyclecounter() - t_kernel;
  if (dbg) printf("bcr_level %d wg %d wave %d (q %d) cycles: load %lld | per step: panel0 %lld  tile %lld  panel1+commit %lld  Z %lld  stores+U %lld  barrier %lld\n",
                  level, bid, wave, q, tph[0], tph[1] / q, tph[2] / q, tph[3] / q, tph[4] / q, tph[5] / q, tph[6] / q);
#undef LTICK
  if (left >= 0) {
    if (role == 0) {
      if (LA ? la_acc_owner : (wave < 4 && !(ELIM && wave == 1))) {
        const int it = LA ? (wave == 6 ? 1 : 0) : wave >> 1, jt = LA ? (wave == 6 ? 1 : 0) : wave & 1;
        double* dst = pendD_w + (size_t(left) * 2 + 1) * BB;
#pragma unroll
        for (int r = 0; r < 4; ++r) put(dst + (16 * it + lk + 4 * r) * BP + 16 * jt + l16, acc_a[r]);
        if (la_acc_owner2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) put(dst + (16 + lk + 4 * r) * BP + l16, acc_a2[r]);
        }
      }
    } else if (LA ? la_acc_owner : (wave == 2 || wave == 3)) {
      const int h = LA ? wave - 5 : wave - 2;
      double* dst = pendF_w + (size_t(left) * 2 + 1) * fblk + f0 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) put(dst + size_t(16 * h + lk + 4 * r) * m1p, acc_a[r]);
    }
  }
  if (CAL_DEV_TIMING(a.debug >= 4 && !LA && tid == 0 && bid < 4 && q > 1)) printf("bcr_level %d wg %d, step 1: loaders (waves 4..7) begin their requests at %.0f %.0f %.0f %.0f, have issued them at %.0f %.0f %.0f %.0f, have committed at %.0f %.0f %.0f %.0f | chief / followers through at %.0f %.0f %.0f %.0f\n", level, bid,
      bcast[20], bcast[21], bcast[22], bcast[23], bcast[28], bcast[29], bcast[30], bcast[31], bcast[12], bcast[13], bcast[14], bcast[15], bcast[8], bcast[9], bcast[10], bcast[11]);
  if (CAL_DEV_TIMING(a.debug >= 4 && LA && tid == 0 && bid < 4 && q > 1)) printf("bcr_level %d wg %d: arrival at the second barrier of step 1 by wave: %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f | end of the elimination (commit for 4..7) of step 1 by wave: %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f | wave 0, step 1: behind the first barrier %.0f, operands there %.0f, products done %.0f\n", level, bid, bcast[0], bcast[1], bcast[2], bcast[3], bcast[4], bcast[5], bcast[6], bcast[7], bcast[8], bcast[9], bcast[10], bcast[11], bcast[12], bcast[13], bcast[14], bcast[15], bcast[16], bcast[17], bcast[18]);
  if (CAL_DEV_TIMING(a.debug >= 4 && tid == 0 && (bid < 9 || bid % 7 == 0))) printf("bcr_level %d: chain workgroup %d (role %d) lived %lld clocks: set-up done at %lld, first block in LDS at %lld, chain done at %lld | steps begin %lld %lld %lld %lld | wave 0 through with its part %lld %lld %lld %lld | Z there %lld %lld %lld %lld\n", level, bid, role, (long long)(__builtin_readcyclecounter() - t_kernel), t_setup, t_first, t_loop, t_top[0], t_top[1], t_top[2], t_top[3], t_elim[0], t_elim[1], t_elim[2], t_elim[3], t_bara[0], t_bara[1], t_bara[2], t_bara[3]);
  if (role == 0 && wave == 0 && lane == 0 && !(pmin > 0.0)) st->chol_failed = 1;
  if (pub) fanin_arrive(fan_word);
}

// ---------------------------------------------------------------------------
// Reduced system in its final indexing [calibration | root | right-hand side]: the calibration / right-hand-side part
// is S - YᵀY over all eliminated rows (one 16x16 tile per workgroup and K-slice, on the matrix cores straight from
// global memory, like schur_kernel), the root part comes from the root superblock (+ its pending updates).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bcr_schur_kernel(SolveArgs a, BcrArgs b, int ks, int n_tile_wg, LmOptionsDev o) {
  const LmState* st = a.st;
  if (st->terminated) return;
  use_current_R(a);
  const FromR fr = {a, o, st->radius};
  if (int(blockIdx.x) >= n_tile_wg) {
    schur_root_rows<false>(a, b, ks, int(blockIdx.x) - n_tile_wg, int(gridDim.x) - n_tile_wg, 256);
    return;
  }
  __shared__ double sacc[4 * 256];
  schur_tile<4, false>(a, b, fr, int(blockIdx.x) / ks, int(blockIdx.x) % ks, ks, sacc, nullptr, 0, -1, -1);
}

// ---------------------------------------------------------------------------
// Back-substitution down the tree + update of the candidate point.
// ---------------------------------------------------------------------------
// Solution of a separator: the root's comes from the reduced solve (behind the calibration part of y), the others'
// from the level above.
DEVI const double* sep_solution(const SolveArgs& a, const BcrArgs& b, int blk) {
  return blk == b.root ? a.y + a.n_s() + a.mc : b.ysol + size_t(blk) * BP;
}

// delta = -y ; candidate = Plus(x, delta) for the parameter blocks selected by the caller; partial sums of the model
// cost change and of the step norms go to the node's slot. (delta / Plus as in update_body, solve_kernels.hip.)
struct UpdSums { double mcc, sn, cn; int bad; };
DEVI void update_block(const BlockDev B, const double* yb, const double* __restrict__ x, double* __restrict__ x_cand, UpdSums& s) {
  const double* p = x + B.amb_off;
  double* qv = x_cand + B.amb_off;
  if (B.manifold == 1) {
    const double d0 = -yb[0], d1 = -yb[1], d2 = -yb[2];
    const double nd = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    double nx = p[0], ny = p[1], nz = p[2], nw = p[3];
    if (nd > 0.0) {
      const double sd = sin(nd) / nd, qw = cos(nd);
      const double qx = sd * d0, qy = sd * d1, qz = sd * d2;
      const double px = p[0], py = p[1], pz = p[2], pw = p[3];
      nw = qw * pw - qx * px - qy * py - qz * pz;
      nx = qw * px + qx * pw + qy * pz - qz * py;
      ny = qw * py + qy * pw + qz * px - qx * pz;
      nz = qw * pz + qz * pw + qx * py - qy * px;
    }
    qv[0] = nx; qv[1] = ny; qv[2] = nz; qv[3] = nw;
    const double e[4] = {p[0] - nx, p[1] - ny, p[2] - nz, p[3] - nw};
    s.sn += e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
    s.cn += nx * nx + ny * ny + nz * nz + nw * nw;
  } else {
    for (int i = 0; i < B.size; ++i) {
      const double v = p[i] - yb[i];
      qv[i] = v; const double e = p[i] - v; s.sn += e * e; s.cn += v * v;
    }
  }
}
// fixed-shape reduction of the per-thread sums of one workgroup into slot `slot`: shuffle tree inside every wave, then
// one thread per quantity adds the waves up in order (one barrier instead of a log-depth LDS tree)
DEVI void file_update_sums(const UpdSums& s, double* sh /* [4][waves] */, double* upd, int slot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  double v[4] = {s.mcc, s.sn, s.cn, s.bad ? 1.0 : 0.0};
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) { sh[wave] = v[0]; sh[16 + wave] = v[1]; sh[32 + wave] = v[2]; sh[48 + wave] = v[3]; }
  // (orders LDS only: __syncthreads() would also wait for the stores of the solutions and the candidate point right in front of
  //  this call to be written through -- a second store drain, ~2k clocks, in front of the one that ends the kernel)
  lds_barrier();
  if (tid < 4) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += sh[16 * tid + w];
    upd[size_t(slot) * 4 + tid] = t;
  }
}

// grid = n_nodes, and for the top level (the first launch after the reduced solve) + 1 workgroup that updates the
// calibration blocks and the root's control points + N·32/8 workgroups that form L⁻¹g - Z^F y_c for the rows of every
// superblock (b.zb), which the levels below read instead of sweeping the border rows again. Dynamic LDS:
// bcr_back_lds_bytes.
constexpr int kBackThreads = 512;
// Update of the calibration blocks (their BlockDevs follow the control points') and of the root's control points,
// whose solution comes from the reduced solve; one workgroup of kBackThreads threads.
template <bool HO>
DEVI void back_calib(const SolveArgs& a, const BcrArgs& b, const double* __restrict__ x, double* __restrict__ x_cand,
                     const BlockDev* __restrict__ blocks, int n_blocks, double* sh, double* ylds, const Handoff& ho) {
  LmState* st = a.st;
  const int tid = threadIdx.x;
  const int n_s = a.n_s(), mc = a.mc;
  constexpr int RB = 6 * kBcrCps;
  UpdSums s = {0.0, 0.0, 0.0, 0};
  {
    // what does not depend on the reduced solve is requested first
    const int jg = min(tid, max(mc - 1, 0));
    const double g_c = a.R[a.off_g() + n_s + jg], d_c = a.dadd[n_s + jg];
    const int tr = RB * max(b.root, 0) + min(tid, RB - 1);
    const bool root_row = b.root >= 0 && tid < RB && tr < n_s;
    const double g_r = a.R[a.off_g() + (root_row ? tr : 0)], d_r = a.dadd[root_row ? tr : 0];
    if (HO) handoff_wait(ho);
    // the solution of the reduced system [calibration | root] into LDS
    const int ny = mc + (b.root >= 0 ? RB : 0);
    for (int j = tid; j < ny; j += kBackThreads) ylds[j] = load_y<HO>(a.y + n_s + j);
    __syncthreads();
    // calibration blocks (their BlockDevs follow the control points') and the root's control points
    if (tid < mc) {
      const double yj = ylds[tid];
      if (!isfinite(yj)) s.bad = 1;
      s.mcc += 0.5 * yj * (g_c + yj * d_c);
    }
    for (int j = tid + kBackThreads; j < mc; j += kBackThreads) {
      const double yj = ylds[j];
      if (!isfinite(yj)) s.bad = 1;
      s.mcc += 0.5 * yj * (a.R[a.off_g() + n_s + j] + yj * a.dadd[n_s + j]);
    }
    for (int bi = tid; bi < n_blocks; bi += kBackThreads) {
      const BlockDev B = blocks[bi];
      if (B.tan_off < n_s) continue;
      update_block(B, ylds + (B.tan_off - n_s), x, x_cand, s);
    }
    if (b.root >= 0) {
      const double* yr = ylds + mc;
      if (root_row) {
        const double yj = yr[tid];
        if (!isfinite(yj)) s.bad = 1;
        s.mcc += 0.5 * yj * (g_r + yj * d_r);
        a.y[tr] = yj;
      }
      if (tid < kBcrCps) {
        const int cp = kBcrCps * b.root + tid;
        const int bi = cp < a.n_cp ? b.cp_block[cp] : -1;
        if (bi >= 0) update_block(blocks[bi], yr + 6 * tid, x, x_cand, s);
      }
    }
    if (tid == 0) { st->rfill = st->rcur ^ 1; st->upd_parts = 0; }
    file_update_sums(s, sh, b.upd, b.n_slots - 1);
  }
}

// One node of the tree: back-substitution of its chain and update of the candidate point of its control points; one
// workgroup of kBackThreads threads. `top`: the node forms L⁻¹g - Z^F y_c itself (sweeping its border rows) instead of
// reading b.zb. QM bounds the unrolled load batches (longest chain of the level).
template <int QM, int MODE, bool HO>      // MODE 0: reads b.zb; 1: `top`; 2: top + the top separators beside the chain (BcrTopSeps)
DEVI void back_node(const SolveArgs& a, const BcrArgs& b, const BcrNodeDev nd, int /*top*/, int q_max, int terminated,
                    bool dbg_first, const double* __restrict__ x, double* __restrict__ x_cand, double* lds, double* sh,
                    const BcrTopSeps& ts, const Handoff& ho) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_s = a.n_s(), mc = a.mc, m1p = b.m1p;
  const size_t fblk = size_t(BP) * m1p;
  constexpr int RB = 6 * kBcrCps;
  constexpr bool SIDE = MODE == 2;
  constexpr int top = MODE >= 1 ? 1 : 0;       // compile-time: the border rows' registers exist only where they are used
  UpdSums s = {0.0, 0.0, 0.0, 0};
  const int q = nd.q, nd_left = nd.left, nd_right = nd.right, nd_slot = nd.slot, blk0 = nd.blk0;
  double* ZBs = lds;                                   // [q][32][33]
  double* Ms = ZBs + size_t(q_max) * BP * DLD;         // [q][32][33]
  double* tv = Ms + size_t(q_max) * BP * DLD;          // [8][32]
  double* ya = tv + kBcrMaxChain * BP;                 // [32]
  double* yn = ya + BP;                                // [32]  solution of the next block (right separator first)
  double* wv = yn + BP;                                // [32]
  double* yc = wv + BP;                                // [m1p]
  double* ych = yc + m1p;                              // [q_max][32] the chain's solutions
  double* wsep = ych + size_t(q_max) * BP;             // [2][32] top separators solved here: L⁻¹g - Z^F y_c - Z^A y_root - Z^B y_root
  double* ysr = wsep + 2 * BP;                         // [32] solution of the top separator on the right (this node files it)
  double* yroot = ysr + BP;                            // [32]
  // Top separators next to this chain (their own launch is gone: see BcrTopSeps). Side 0 = left, 1 = right; the node
  // that has one on its right files its solution and updates its control points (threads 480.. and 448..).
  int sk[2] = {-1, -1};
  if (SIDE) for (int k = 0; k < ts.n; ++k) { if (nd_left == ts.blk[k]) sk[0] = k; if (nd_right == ts.blk[k]) sk[1] = k; }
  const bool side = SIDE && (sk[0] >= 0 || sk[1] >= 0);
  const int sep_r = sk[1] >= 0 ? nd_right : -1;
  // Everything the node needs is requested before anything is consumed -- the separators' solutions, Z^B and L⁻ᵀ of
  // every block (to LDS), this thread's entries of Z^A and of L⁻¹g - Z^F y_c (sixteen threads per row), and what the
  // update of the candidate point reads (gradient, damping, the control points' current values): a dependent global
  // load costs about a microsecond here, the arithmetic next to nothing. Chain indices are clamped, not predicated,
  // so that the loads stay unconditional.
  const bool bdbg = CAL_DEV_TIMING(a.debug == 1 && dbg_first && tid == 0);
  long long bt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, btk = bdbg ? __builtin_readcyclecounter() : 0;
#define BTICK(i) if (bdbg) { const long long t_ = __builtin_readcyclecounter(); bt[i] += t_ - btk; btk = t_; }
  // update stage: thread e < 32q owns row e of the chain (gradient, damping), thread e < 5q control point e. The offset
  // of the control point goes first: its values are a dependent load, and loads return in order.
  const bool sep_row = sep_r >= 0 && tid >= 480, sep_cp = sep_r >= 0 && tid >= 448 && tid < 448 + kBcrCps;
  const int my_row_t = sep_row ? RB * sep_r + (tid & 31) : RB * (blk0 + (tid >> 5)) + (tid & 31);
  const bool my_row_ok = (tid < q * BP || sep_row) && (tid & 31) < RB && my_row_t < n_s;
  const int my_cp = sep_cp ? kBcrCps * sep_r + (tid - 448) : kBcrCps * blk0 + tid;     // the chain's control points are consecutive
  const bool my_cp_in = (tid < q * kBcrCps || sep_cp) && my_cp < a.n_cp;
  const int my_cp_c = my_cp_in ? my_cp : 0;
  const int my_off = b.ctrl_off[my_cp_c];
  // (HO: the reduced system's solution is requested behind the hand-off, further down; everything else goes first)
  double ysep = 0.0, ycv = 0.0;
  auto load_solution = [&]() {
    const double* pl = nd_left >= 0 && sk[0] < 0 ? sep_solution(a, b, nd_left) : a.y;
    const double* pr = nd_right >= 0 && sk[1] < 0 ? sep_solution(a, b, nd_right) : a.y;
    const double vl = load_y<HO>(pl + (tid & 31)), vr = load_y<HO>(pr + (tid & 31));
    // the root's solution comes from the reduced solve, which knows its 30 real rows only: the two padding rows are 0,
    // not whatever sits behind them in y (an uninitialised word there may be a NaN, and NaN times a zero column is NaN)
    const bool pad = (tid & 31) >= RB;
    ysep = tid < BP ? ((nd_left >= 0 && !(pad && nd_left == b.root)) ? vl : 0.0)
                    : ((nd_right >= 0 && !(pad && nd_right == b.root)) ? vr : 0.0);
    ycv = load_y<HO>(a.y + n_s + min(tid, mc - 1 > 0 ? mc - 1 : 0));
  };
  if (!HO) load_solution();
  const int r16 = tid >> 4, sub = tid & 15;
  // (16-byte loads throughout: a workgroup issues a 64-lane load every ~12 clocks whatever its width, and the node waits
  //  for a few hundred of them. Thread (r16, sub) owns columns 2·sub, 2·sub + 1 (+ 32u for the border) of row r16.)
  double2 s_yv[2][4], s_za[2], s_zb[2], s_mm[2];
  double s_zt[2], yroot_v = 0.0;
  if (side) {
    // (the root's solution: its 30 real rows, see above)
    if (!HO) yroot_v = (b.root >= 0 && (tid & 31) < RB) ? a.y[n_s + mc + (tid & 31)] : 0.0;
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      const int eb = sk[sd] >= 0 ? ts.blk[sk[sd]] : blk0;       // (clamped, not predicated: harmless loads)
      const double* yrow = b.Y + size_t(eb) * fblk + size_t(r16) * m1p;
      s_zt[sd] = yrow[mc];
#pragma unroll
      for (int u = 0; u < 4; ++u) s_yv[sd][u] = reinterpret_cast<const double2*>(yrow)[min(sub + 16 * u, m1p / 2 - 1)];
      const size_t g2 = size_t(eb) * (BB / 2) + tid;
      s_za[sd] = reinterpret_cast<const double2*>(b.ZA)[g2];
      s_zb[sd] = reinterpret_cast<const double2*>(b.ZB)[g2];
      s_mm[sd] = reinterpret_cast<const double2*>(b.M)[g2];
    }
  }
  static_assert(kBackThreads * 2 == BB, "one 16-byte load per thread covers a 32x32 block");
  double2 vz[QM], vm[QM], za[QM], yv[QM][4];
  double zt[QM];
#pragma unroll
  for (int i = 0; i < QM; ++i) {
    const int blk = blk0 + min(i, q - 1);
    const size_t g2 = size_t(blk) * (BB / 2) + tid;
    vz[i] = reinterpret_cast<const double2*>(b.ZB)[g2];
    vm[i] = reinterpret_cast<const double2*>(b.M)[g2];
    za[i] = reinterpret_cast<const double2*>(b.ZA)[g2];
    const double* yrow = b.Y + size_t(blk) * fblk + size_t(r16) * m1p;
    zt[i] = top ? yrow[mc] : b.zb[size_t(blk) * BP + r16];
    if (top) {       // this thread's share of the border row (the first 128 calibration columns)
#pragma unroll
      for (int u = 0; u < 4; ++u) yv[i][u] = reinterpret_cast<const double2*>(yrow)[min(sub + 16 * u, m1p / 2 - 1)];
    }
  }
  const double my_g = a.R[a.off_g() + (my_row_ok ? my_row_t : 0)], my_dadd = a.dadd[my_row_ok ? my_row_t : 0];
  const bool my_cp_ok = my_cp_in && (b.all_active || a.cp_active[my_cp_c] != 0);
  double px[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) px[c] = x[my_off + c];
  if (terminated) return;
  if (HO) {
    // everything above is on its way (or here) while the reduced solve is still running in workgroup 0 of this launch;
    // what does not depend on its solution goes to LDS before the wait, too
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      if (i < q) {
        const int o = (i * BP + r16) * DLD + 2 * sub;
        ZBs[o] = vz[i].x; ZBs[o + 1] = vz[i].y; Ms[o] = vm[i].x; Ms[o + 1] = vm[i].y;
      }
    }
    handoff_wait(ho);
    load_solution();
    if (side) yroot_v = (b.root >= 0 && (tid & 31) < RB) ? load_sc1(a.y + n_s + mc + (tid & 31)) : 0.0;
  }
  BTICK(0)
  if (CAL_DEV_TIMING(a.debug == 2)) {     // development aid: which input of the node is not finite?
    bool bz = false, bm = false, ba = false, bt = false;
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      if (i < q) { bz = bz || !isfinite(vz[i].x) || !isfinite(vz[i].y); bm = bm || !isfinite(vm[i].x) || !isfinite(vm[i].y);
                   ba = ba || !isfinite(za[i].x) || !isfinite(za[i].y); bt = bt || !isfinite(zt[i]); }
    }
    if (bz || bm || ba || bt || (tid < 2 * BP && !isfinite(ysep)) || (top && tid < mc && !isfinite(ycv)))
      printf("bcr_back top %d node %d (blk0 %d q %d left %d right %d) tid %d: ZB %d M %d ZA %d zt %d ysep %d ycv %d\n", top, int(blockIdx.x), blk0, q,
             nd_left, nd_right, tid, int(bz), int(bm), int(ba), int(bt), int(tid < 2 * BP && !isfinite(ysep)), int(top && tid < mc && !isfinite(ycv)));
  }
  if (tid < BP) ya[tid] = ysep; else if (tid < 2 * BP) yn[tid - BP] = ysep;
  if (side && tid < BP) yroot[tid] = yroot_v;
  if (top) {
    if (tid < m1p) yc[tid] = tid < mc ? ycv : 0.0;
    for (int j = tid + kBackThreads; j < m1p; j += kBackThreads) yc[j] = j < mc ? load_y<HO>(a.y + n_s + j) : 0.0;
  }
  if (!HO) {
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      if (i < q) {
        const int o = (i * BP + r16) * DLD + 2 * sub;
        ZBs[o] = vz[i].x; ZBs[o + 1] = vz[i].y; Ms[o] = vm[i].x; Ms[o + 1] = vm[i].y;
      }
    }
  }
  __syncthreads();
  // this thread's columns of y_c (2·sub + 32u, + 1), read once for all blocks: 16-byte LDS reads, zero from column mc on
  double ycx[4], ycy[4];
  if (top) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 2 * sub + 32 * u;
      const double2 y2 = *reinterpret_cast<const double2*>(yc + min(c, m1p - 2));
      ycx[u] = c < mc ? y2.x : 0.0; ycy[u] = c + 1 < mc ? y2.y : 0.0;
    }
  }
  if (side) {
    // y_e = L⁻ᵀ(L⁻¹g_e - Z^F y_c - Z^A y_root - Z^B y_root) of the top separators on either side (sixteen threads per row),
    // both sides at once (the loads of a side that does not exist were clamped to valid memory: its result is dropped)
    const double2 yr2 = *reinterpret_cast<const double2*>(yroot + 2 * sub);
    double part[2];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      const int k = max(sk[sd], 0);
      double pt = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) pt += s_yv[sd][u].x * ycx[u] + s_yv[sd][u].y * ycy[u];
      if (mc > 128 && sk[sd] >= 0) {
        const double* yrow = b.Y + size_t(ts.blk[k]) * fblk + size_t(r16) * m1p;
        for (int j = sub + 128; j < mc; j += 16) pt += yrow[j] * yc[j];
      }
      pt += ts.left[k] >= 0 ? s_za[sd].x * yr2.x + s_za[sd].y * yr2.y : 0.0;
      pt += ts.right[k] >= 0 ? s_zb[sd].x * yr2.x + s_zb[sd].y * yr2.y : 0.0;
      part[sd] = pt;
    }
    part[0] = row16_sum(part[0]); part[1] = row16_sum(part[1]);
    if (sub == 0) { wsep[r16] = s_zt[0] - part[0]; wsep[BP + r16] = s_zt[1] - part[1]; }
    __syncthreads();
    double yp[2];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      const double2 w2 = *reinterpret_cast<const double2*>(wsep + sd * BP + 2 * sub);
      yp[sd] = s_mm[sd].x * w2.x + s_mm[sd].y * w2.y;
    }
    yp[0] = row16_sum(yp[0]); yp[1] = row16_sum(yp[1]);
    if (sub == 0) {
      if (sk[0] >= 0) ya[r16] = yp[0];
      if (sk[1] >= 0) { yn[r16] = yp[1]; ysr[r16] = yp[1]; }
    }
    __syncthreads();
  }
  BTICK(1)
  // t_i = (L⁻¹g_i - Z^F y_c) - Z^A y_a : sixteen threads per row; all blocks' partial sums first, then their reductions
  // side by side (independent chains), then the stores -- block by block behind branches each step waited for the last
  {
    double2 ya2 = {0.0, 0.0};
    if (nd_left >= 0) ya2 = *reinterpret_cast<const double2*>(ya + 2 * sub);
    double part[QM];
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      double pt = za[i].x * ya2.x + za[i].y * ya2.y;
      if (top) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pt += yv[i][u].x * ycx[u] + yv[i][u].y * ycy[u];
        if (mc > 128 && i < q) {
          const double* yrow = b.Y + size_t(blk0 + i) * fblk + size_t(r16) * m1p;
          for (int j = sub + 128; j < mc; j += 16) pt += yrow[j] * yc[j];
        }
      }
      part[i] = pt;
    }
#pragma unroll
    for (int i = 0; i < QM; ++i) part[i] = row16_sum(part[i]);
#pragma unroll
    for (int i = 0; i < QM; ++i) if (i < q && sub == 0) tv[i * BP + r16] = zt[i] - part[i];
  }
  __syncthreads();
  BTICK(2)
  // the chain, last block first: w = t_i - Z^B y_next ; y_i = L⁻ᵀ w (one wave, two lanes per row)
  if (wave == 0) {
    const int r = lane & 31, h = lane >> 5;
    for (int i = q - 1; i >= 0; --i) {
      const double* zb = ZBs + (i * BP + r) * DLD + 16 * h;
      double wp = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) wp += zb[j] * yn[16 * h + j];
      wp += other_half(wp);
      const double w = tv[i * BP + r] - wp;
      if (h == 0) wv[r] = w;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double* mr = Ms + (i * BP + r) * DLD + 16 * h;
      double yp = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) yp += mr[c] * wv[16 * h + c];
      yp += other_half(yp);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (h == 0) { yn[r] = yp; ych[i * BP + r] = yp; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  BTICK(3)
  __syncthreads();
  BTICK(4)
  // file the solutions, update the candidate point of the chain's control points (delta = -y, plain vector blocks)
  if (tid < q * BP || sep_row) {
    const double yj = sep_row ? ysr[tid & 31] : ych[tid];
    b.ysol[sep_row ? size_t(sep_r) * BP + (tid & 31) : size_t(blk0) * BP + tid] = yj;
    if (my_row_ok) {
      if (!isfinite(yj)) s.bad = 1;
      s.mcc += 0.5 * yj * (my_g + yj * my_dadd);
      a.y[my_row_t] = yj;
    }
  }
  if (my_cp_ok) {
    const double* yb = sep_cp ? ysr + 6 * (tid - 448) : ych + (tid / kBcrCps) * BP + 6 * (tid % kBcrCps);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double v = px[c] - yb[c];
      x_cand[my_off + c] = v;
      const double e = px[c] - v;
      s.sn += e * e; s.cn += v * v;
    }
  }
  BTICK(5)
  file_update_sums(s, sh, b.upd, nd_slot);
  BTICK(6)
  if (bdbg) printf("bcr_back top %d (q %d) cycles: loads issued+arrived %lld  to LDS+barrier %lld  t-phase %lld  chain %lld  barrier %lld  outputs %lld  sums %lld\n",
                   top, q, bt[0], bt[1], bt[2], bt[3], bt[4], bt[5], bt[6]);
#undef BTICK
}

// ---------------------------------------------------------------------------
// A node riding in the dense solve's launch, with its work moved IN FRONT of the hand-off. Everything a node computes is
// affine in what the reduced solve hands over, u = [y_c | y_root | 1]:
//     top separator s beside the chain:  y_s = M_s (z_s - Y_s y_c - Z^A_s y_root - Z^B_s y_root)
//     chain block i (last first):        y_i = M_i (z_i - Y_i y_c - Z^A_i y_a - Z^B_i y_next)
// so while the solve is running the node forms X with y = X u block by block -- X_i = M_i ([-Y_i | 0 | z_i] - Z^A_i X_a -
// Z^B_i X_next), products of 32x32 blocks with 32 x 128 tiles on the matrix cores -- and behind the hand-off all that is
// left is one product with u: the t-phase, the top separators' solves and the chain of dependent 32x32 matrix-vector
// products (~14k clocks behind the hand-off in back_node) become ~4k.
// Wave w owns columns [16w, 16w + 16) of every X through the whole recursion: the MFMA result layout (row = lk + 4r,
// column = l16) of one step IS the B-operand layout (k = lk + 4u, column = l16) of the next, so the tiles never leave
// the wave's registers and the recursion needs no barrier. NCOL = mc + 33 <= 128 (the host checks).
// ---------------------------------------------------------------------------
DEVI void mat_times_tile(const double* Z /* LDS [32][DLD] */, const f64x4& b0, const f64x4& b1, f64x4& d0, f64x4& d1, int l16, int lk, bool neg) {
  double a0[8], a1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { a0[u] = Z[l16 * DLD + lk + 4 * u]; a1[u] = Z[(16 + l16) * DLD + lk + 4 * u]; }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double bv = u < 4 ? b0[u & 3] : b1[u & 3];
    d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(neg ? -a0[u] : a0[u], bv, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(neg ? -a1[u] : a1[u], bv, d1, 0, 0, 0);
  }
}
// D = Z·B as above AND Dᵗ-tiles of the same product for the final product with u: (Z B)ᵀ = Bᵀ Zᵀ, whose A operand is B in
// the MFMA RESULT layout (in-lane again) and whose B operand is the very values read for Z above. dt0 / dt1: rows 0..15 /
// 16..31 of D along the lanes (l16), this wave's columns lk + 4r along the registers.
DEVI void mat_times_tile_both(const double* Z, const f64x4& b0, const f64x4& b1, f64x4& d0, f64x4& d1, f64x4& dt0, f64x4& dt1, int l16, int lk) {
  double a0[8], a1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { a0[u] = Z[l16 * DLD + lk + 4 * u]; a1[u] = Z[(16 + l16) * DLD + lk + 4 * u]; }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double bv = u < 4 ? b0[u & 3] : b1[u & 3];
    d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], bv, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], bv, d1, 0, 0, 0);
    dt0 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, a0[u], dt0, 0, 0, 0);
    dt1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, a1[u], dt1, 0, 0, 0);
  }
}
size_t bcr_back_pre_lds_doubles(int q_max) { return size_t(3 * q_max + 2) * BP * DLD + 128 + size_t(8) * (q_max + 1) * BP + size_t(q_max + 1) * BP; }
template <int QM, bool SIDE>
DEVI void back_node_pre(const SolveArgs& a, const BcrArgs& b, const BcrNodeDev nd, int terminated,
                        const double* __restrict__ x, double* __restrict__ x_cand, double* lds, double* sh,
                        const BcrTopSeps& ts, const Handoff& ho) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lk = lane >> 4;
  const int n_s = a.n_s(), mc = a.mc, m1p = b.m1p;
  const size_t fblk = size_t(BP) * m1p;
  constexpr int RB = 6 * kBcrCps;
  UpdSums s = {0.0, 0.0, 0.0, 0};
  const int q = nd.q, nd_left = nd.left, nd_right = nd.right, nd_slot = nd.slot, blk0 = nd.blk0;
  double* ZAs = lds;                                   // [QM][32][33]
  double* ZBs = ZAs + size_t(QM) * BP * DLD;           // [QM][32][33]
  double* Ms = ZBs + size_t(QM) * BP * DLD;            // [QM][32][33]
  double* Msep = Ms + size_t(QM) * BP * DLD;           // [2][32][33]  L⁻ᵀ of the top separators beside the chain
  double* uv = Msep + 2 * BP * DLD;                    // [128] u = [y_c | y_root | 1 | 0...]
  double* part = uv + 128;                             // [8][QM + 1][32] per-wave partial products
  double* ych = part + 8 * (QM + 1) * BP;              // [QM][32] the chain's solutions, then [32] the right top separator's
  double* ysr = ych + QM * BP;
  int sk[2] = {-1, -1};
  if (SIDE) for (int k = 0; k < ts.n; ++k) { if (nd_left == ts.blk[k]) sk[0] = k; if (nd_right == ts.blk[k]) sk[1] = k; }
  const int sep_r = sk[1] >= 0 ? nd_right : -1;
  // update stage (as in back_node): thread e < 32q owns row e of the chain, thread e < 5q control point e; the node that has
  // a top separator on its right files that one's solution, too (threads 480.. and 448..)
  const bool sep_row = sep_r >= 0 && tid >= 480, sep_cp = sep_r >= 0 && tid >= 448 && tid < 448 + kBcrCps;
  const int my_row_t = sep_row ? RB * sep_r + (tid & 31) : RB * (blk0 + (tid >> 5)) + (tid & 31);
  const bool my_row_ok = (tid < q * BP || sep_row) && (tid & 31) < RB && my_row_t < n_s;
  const int my_cp = sep_cp ? kBcrCps * sep_r + (tid - 448) : kBcrCps * blk0 + tid;
  const bool my_cp_in = (tid < q * kBcrCps || sep_cp) && my_cp < a.n_cp;
  const int my_cp_c = my_cp_in ? my_cp : 0;
  const int my_off = b.ctrl_off[my_cp_c];
  const bool pdbg = CAL_DEV_TIMING(a.debug == 1 && blk0 == 0 && tid == 0);
  long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptk = pdbg ? __builtin_readcyclecounter() : 0;
#define PTICK(i) if (pdbg) { const long long t_ = __builtin_readcyclecounter(); pt[i] += t_ - ptk; ptk = t_; }
  // ---- requests: operands for LDS (thread (r16, sub): two entries of a row) ----
  const int r16 = tid >> 4, sub = tid & 15;
  double2 vza[QM], vzb[QM], vm[QM], vms[2];
#pragma unroll
  for (int i = 0; i < QM; ++i) {
    const size_t g2 = size_t(blk0 + min(i, q - 1)) * (BB / 2) + tid;
    vza[i] = reinterpret_cast<const double2*>(b.ZA)[g2];
    vzb[i] = reinterpret_cast<const double2*>(b.ZB)[g2];
    vm[i] = reinterpret_cast<const double2*>(b.M)[g2];
  }
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    const int eb = SIDE && sk[sd] >= 0 ? ts.blk[sk[sd]] : blk0;
    vms[sd] = reinterpret_cast<const double2*>(b.M)[size_t(eb) * (BB / 2) + tid];
  }
  // ---- requests: right-hand-side tiles in the MFMA result layout: rows lk + 4r (+ 16), column 16·wave + l16 of
  //      [-Y | 0 | z] (chain) and [-Y | -Z^A - Z^B | z] (top separators) ----
  const int col = 16 * wave + l16;
  const int cy = min(col, mc);                          // column of Y to read (mc: the right-hand side z = L⁻¹g)
  const bool is_y = col < mc, is_z = col == mc + BP, is_r = col >= mc && col < mc + BP;
  const int cr = min(max(col - mc, 0), BP - 1);
  // (wave-uniform: a wave requests only what one of its sixteen columns selects below -- the requests of the eight waves of
  //  a node, not their round trip, are what this phase takes)
  const bool w_yz = uniform(int(16 * wave < mc || ((mc + BP) >> 4) == wave)) != 0;          // a column of Y, or z
  const bool w_r = uniform(int(16 * wave + 15 >= mc && 16 * wave < mc + BP)) != 0;          // a column of the root part
  double rh[QM][8], ws[2][8];
#pragma unroll
  for (int i = 0; i < QM; ++i) {
    const double* yb = b.Y + size_t(blk0 + min(i, q - 1)) * fblk;
#pragma unroll
    for (int e = 0; e < 8; ++e) rh[i][e] = 0.0;
    if (w_yz) {
#pragma unroll
      for (int e = 0; e < 8; ++e) rh[i][e] = yb[size_t(lk + 4 * (e & 3) + 16 * (e >> 2)) * m1p + (is_z ? mc : cy)];
    }
  }
  if (SIDE) {
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      const int k = max(sk[sd], 0);
      const int eb = sk[sd] >= 0 ? ts.blk[k] : blk0;
      const double* yb = b.Y + size_t(eb) * fblk;
      const double* za = b.ZA + size_t(eb) * BB;
      const double* zb = b.ZB + size_t(eb) * BB;
      double vy[8], va[8], vb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { vy[e] = 0.0; va[e] = 0.0; vb[e] = 0.0; }
      if (w_yz) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vy[e] = yb[size_t(lk + 4 * (e & 3) + 16 * (e >> 2)) * m1p + (is_z ? mc : cy)];
      }
      if (w_r) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = lk + 4 * (e & 3) + 16 * (e >> 2);
          va[e] = za[row * BP + cr]; vb[e] = zb[row * BP + cr];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const double vr = (ts.left[k] >= 0 ? va[e] : 0.0) + (ts.right[k] >= 0 ? vb[e] : 0.0);
        ws[sd][e] = is_y ? -vy[e] : (is_r ? -vr : (is_z ? vy[e] : 0.0));
      }
    }
  }
  const double my_g = a.R[a.off_g() + (my_row_ok ? my_row_t : 0)], my_dadd = a.dadd[my_row_ok ? my_row_t : 0];
  const bool my_cp_ok = my_cp_in && (b.all_active || a.cp_active[my_cp_c] != 0);
  double px[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) px[c] = x[my_off + c];
  if (terminated) return;
  PTICK(0)
  // ---- operands into LDS ----
#pragma unroll
  for (int i = 0; i < QM; ++i) {
    if (i < q) {
      const int o = (i * BP + r16) * DLD + 2 * sub;
      ZAs[o] = vza[i].x; ZAs[o + 1] = vza[i].y; ZBs[o] = vzb[i].x; ZBs[o + 1] = vzb[i].y; Ms[o] = vm[i].x; Ms[o + 1] = vm[i].y;
    }
  }
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) { const int o = (sd * BP + r16) * DLD + 2 * sub; Msep[o] = vms[sd].x; Msep[o + 1] = vms[sd].y; }
  __syncthreads();
  PTICK(1)
  // ---- the recursion, per wave on its column tile ----
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  f64x4 xr0, xr1;          // y_root = u[mc .. mc + 30): rows lk + 4r (+16), column `col`
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row0 = lk + 4 * r, row1 = 16 + lk + 4 * r;
    xr0[r] = (col == mc + row0 && row0 < RB) ? 1.0 : 0.0;
    xr1[r] = (col == mc + row1 && row1 < RB) ? 1.0 : 0.0;
  }
  // (a wave whose 16 columns lie beyond u -- a small calibration part -- has nothing to do: all of its tiles are zero)
  const bool act = 16 * wave < mc + BP + 1;
  f64x4 xs0[2] = {zero4, zero4}, xs1[2] = {zero4, zero4};
  f64x4 XT0[QM + 1], XT1[QM + 1];          // transposed tiles of every block's map (QM: the right top separator's)
#pragma unroll
  for (int i = 0; i <= QM; ++i) { XT0[i] = zero4; XT1[i] = zero4; }
  if (SIDE && act) {
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      if (sk[sd] >= 0) {
        const f64x4 w0 = {ws[sd][0], ws[sd][1], ws[sd][2], ws[sd][3]}, w1 = {ws[sd][4], ws[sd][5], ws[sd][6], ws[sd][7]};
        if (sd == 1) mat_times_tile_both(Msep + sd * BP * DLD, w0, w1, xs0[sd], xs1[sd], XT0[QM], XT1[QM], l16, lk);
        else mat_times_tile(Msep + sd * BP * DLD, w0, w1, xs0[sd], xs1[sd], l16, lk, false);
      }
    }
  }
  const bool has_a = nd_left >= 0, has_n = nd_right >= 0;
  const f64x4 xa0 = sk[0] >= 0 ? xs0[0] : (has_a ? xr0 : zero4), xa1 = sk[0] >= 0 ? xs1[0] : (has_a ? xr1 : zero4);
  f64x4 xn0 = sk[1] >= 0 ? xs0[1] : (has_n ? xr0 : zero4), xn1 = sk[1] >= 0 ? xs1[1] : (has_n ? xr1 : zero4);
#pragma unroll
  for (int ii = 0; ii < QM; ++ii) {
    const int i = QM - 1 - ii;
    if (i < q && act) {
      f64x4 t0, t1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t0[r] = is_y ? -rh[i][r] : (is_z ? rh[i][r] : 0.0);
        t1[r] = is_y ? -rh[i][4 + r] : (is_z ? rh[i][4 + r] : 0.0);
      }
      if (has_a) mat_times_tile(ZAs + i * BP * DLD, xa0, xa1, t0, t1, l16, lk, true);
      if (i + 1 < q || has_n) mat_times_tile(ZBs + i * BP * DLD, xn0, xn1, t0, t1, l16, lk, true);
      f64x4 x0 = zero4, x1 = zero4;
      mat_times_tile_both(Ms + i * BP * DLD, t0, t1, x0, x1, XT0[i], XT1[i], l16, lk);
      xn0 = x0; xn1 = x1;
    }
  }
  PTICK(2)
  // ---- hand-off: u, then one product with it ----
  handoff_wait(ho);
  PTICK(3)
  {
    // this lane's four entries of u: columns 16·wave + lk + 4r (y_c and the root's 30 real rows sit back to back in y)
    double uj[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * wave + lk + 4 * r;
      const double v = load_sc1(a.y + n_s + min(j, mc + RB - 1));
      uj[r] = j < mc ? v : ((j < mc + RB && b.root >= 0) ? v : (j == mc + BP ? 1.0 : 0.0));
    }
    if (pdbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PTICK(5) }
    // rows 16·ct + l16 of every block: four products in the lane, then the four lane groups (lk) added up
    double pv[QM + 1][2];
#pragma unroll
    for (int i = 0; i <= QM; ++i) {
      pv[i][0] = (XT0[i][0] * uj[0] + XT0[i][1] * uj[1]) + (XT0[i][2] * uj[2] + XT0[i][3] * uj[3]);
      pv[i][1] = (XT1[i][0] * uj[0] + XT1[i][1] * uj[1]) + (XT1[i][2] * uj[2] + XT1[i][3] * uj[3]);
    }
#pragma unroll
    for (int i = 0; i <= QM; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) { pv[i][c] += other_half(pv[i][c]); pv[i][c] += other_row(pv[i][c]); }
    if (lk == 0) {
#pragma unroll
      for (int i = 0; i <= QM; ++i) { part[(wave * (QM + 1) + i) * BP + l16] = pv[i][0]; part[(wave * (QM + 1) + i) * BP + 16 + l16] = pv[i][1]; }
    }
  }
  __syncthreads();
  PTICK(6)
  if (tid < (QM + 1) * BP) {
    const int i = tid >> 5, r = tid & 31;
    double y = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) y += part[(w * (QM + 1) + i) * BP + r];
    ych[i * BP + r] = y;      // (block QM: the right top separator = ysr)
  }
  __syncthreads();
  PTICK(7)
  // ---- file the solutions, update the candidate point (as back_node) ----
  if (tid < q * BP || sep_row) {
    const double yj = sep_row ? ysr[tid & 31] : ych[tid];
    b.ysol[sep_row ? size_t(sep_r) * BP + (tid & 31) : size_t(blk0) * BP + tid] = yj;
    if (my_row_ok) {
      if (!isfinite(yj)) s.bad = 1;
      s.mcc += 0.5 * yj * (my_g + yj * my_dadd);
      a.y[my_row_t] = yj;
    }
  }
  if (my_cp_ok) {
    const double* yb = sep_cp ? ysr + 6 * (tid - 448) : ych + (tid / kBcrCps) * BP + 6 * (tid % kBcrCps);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double v = px[c] - yb[c];
      x_cand[my_off + c] = v;
      const double e = px[c] - v;
      s.sn += e * e; s.cn += v * v;
    }
  }
  PTICK(8)
  file_update_sums(s, sh, b.upd, nd_slot);
  PTICK(4)
  if (pdbg) printf("back_node_pre (q %d) cycles: requests issued + arrived %lld  to LDS %lld  recursion %lld  wait for the solve %lld | u there %lld  products + partials in LDS %lld  sums %lld  outputs stored %lld  update sums filed %lld\n",
                   q, pt[0], pt[1], pt[2], pt[3], pt[5], pt[6], pt[7], pt[8], pt[4]);
#undef PTICK
}

// grid = n_nodes; with `extras` (the first launch after the reduced solve when that kernel does not take the top
// level along) + 1 workgroup for back_calib + N·32/8 workgroups that form L⁻¹g - Z^F y_c for the rows of every
// superblock (b.zb), which nodes launched with top = 0 read instead of sweeping the border rows again.
template <int QM, int MODE, bool HO, bool PRE = false>     // longest chain of the level; MODE: see back_node; HO: rides in the dense solve's launch; PRE: back_node_pre
DEVI void bcr_back_body(SolveArgs a, const BcrArgs& b, int wg, int node0, int n_nodes, int top, int q_max,
                        const double* __restrict__ x, double* __restrict__ x_cand,
                        const BlockDev* __restrict__ blocks, int n_blocks, const BcrTopSeps& ts, double* lds, double* sh, const Handoff& ho, int q0 = 0) {
  LmState* st = a.st;
  const int terminated = st->terminated;     // tested after the loads are on their way
  use_current_R(a);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_s = a.n_s(), mc = a.mc, m1p = b.m1p;
  if (wg >= n_nodes && terminated) return;
  if (wg > n_nodes) {
    // z - Z^F y_c, one wave per row of Y   (never part of a launch with a hand-off: see launch_dense_back)
    const int row = (wg - n_nodes - 1) * (kBackThreads / 64) + wave;
    if (row >= b.N * BP) return;
    const double* yrow = b.Y + size_t(row) * m1p;
    const double* yc = a.y + n_s;
    double part = 0.0;
#pragma unroll 2
    for (int j = lane; j < mc; j += 64) part += yrow[j] * yc[j];
    part = wave_sum(part);
    if (lane == 0) b.zb[row] = yrow[mc] - part;
    return;
  }
  if (wg == n_nodes) { back_calib<HO>(a, b, x, x_cand, blocks, n_blocks, sh, lds, ho); return; }
  // (q0 > 0: the nodes are level 0's, which are regular -- [chain of q0] [separator] [chain] ... in time order, slot = number --,
  //  so the descriptor is arithmetic on the node's number: no load in front of the node's requests, one dependent round trip
  //  (~1.9 us behind a kernel boundary) less at the head of the dense solve's launch. The host's table says the same.)
  BcrNodeDev nd;
  if (q0 > 0) {
    nd.blk0 = wg * (q0 + 1); nd.q = min(q0, b.N - nd.blk0);
    nd.left = wg > 0 ? nd.blk0 - 1 : -1; nd.right = nd.blk0 + nd.q < b.N ? nd.blk0 + nd.q : -1; nd.slot = node0 + wg; nd.pend = 0;
  } else nd = b.nodes[node0 + wg];
  if (PRE) back_node_pre<QM, MODE == 2>(a, b, nd, terminated, x, x_cand, lds, sh, ts, ho);
  else back_node<QM, MODE, HO>(a, b, nd, top, q_max, terminated, wg == 0, x, x_cand, lds, sh, ts, ho);
}
template <int QM, int MODE>
__global__ __launch_bounds__(kBackThreads) void bcr_back_kernel(SolveArgs a, BcrArgs b, int node0, int n_nodes, int top, int extras, int q_max,
                                                                const double* __restrict__ x, double* __restrict__ x_cand,
                                                                const BlockDev* __restrict__ blocks, int n_blocks, BcrTopSeps ts) {
  extern __shared__ double lds[];
  __shared__ double sh[64];
  (void)extras;
  bcr_back_body<QM, MODE, false>(a, b, int(blockIdx.x), node0, n_nodes, top, q_max, x, x_cand, blocks, n_blocks, ts, lds, sh, Handoff{nullptr, 0});
}

// ---------------------------------------------------------------------------
// Dense reduced solve for m + 1 <= 128, by 32-column blocks: the same building blocks as a tree node -- D_jj = L Lᵀ with
// L⁻ᵀ riding along as 32 identity rows (two in-wave 16-column panels on 64 rows, one lane per row), the rows below as
// Z = A_ij L⁻ᵀ on the matrix cores, the trailing update as 16x16 MFMA tiles over all eight waves, and a
// back-substitution that is four 32x32 matrix-vector products with the stored L⁻ᵀ instead of a 122-step chain.
// (reduced_solve_panel_kernel factors 16-column panels over ALL rows in one wave, two rows per lane: 8 panels of
// ≈4.9k clocks + 2.0k of column update each and a 17k backward sweep, 74k clocks; here the chain is 8 panels of ≈4k
// on 64 rows and everything else is spread over the workgroup.)
// The matrix lives in LDS (row stride 129: conflict-free column walks), the right-hand side as a separate row; L⁻ᵀ of
// block j is kept in the (otherwise unused) strict upper triangle of the diagonal block, its diagonal in dinvm.
// ---------------------------------------------------------------------------
constexpr int kDenseThreads = 512;
constexpr int DNL = 129;     // row stride of the dense matrix in LDS
constexpr int kDenseChan = 2 * kElimCompactDoubles;      // the panel buffer's place: [64][DLD] or the rolling form's two channels
static_assert(kDenseChan >= 64 * DLD, "the panel buffer must fit where the channels sit");
// (Taking the top level of the tree along in this workgroup -- back_calib + back_node for its one to three nodes -- was
//  tried and lost: two nodes one after the other cost 12 us of dependent loads here against the 7 us of a launch that
//  runs them side by side, and the levels below then sweep their own border rows.)
// t0 > 0: the system is what the blocked multi-launch factorisation (reduced_block_step_kernel) left of a larger one --
// unknowns t0 .. a.m - 1, in place in slice 0 of a.Spart with the full system's row stride, right-hand side in its row a.m.
// elim (round 4): every 32-column block is eliminated by block_elim.hpp in place -- wave 0 the chief on the diagonal block (L
// into the lower triangle), waves 1..3 the followers: the identity rows (L⁻ᵀ: strict upper triangle + dinvm), the row tiles
// below (Z = A_ij L⁻ᵀ comes out of the factorisation, in place) and the right-hand side as a single-row tile (z = g L⁻ᵀ) --,
// waves 4..7 beside them the trailing tiles of the block before that the current block does not touch; between two
// blocks one phase: right-hand side of the rows below, and the trailing update of the NEXT block's two column tiles.
// COH (the body rides in reduced_fused_kernel behind the blocked steps of the same launch): what those steps wrote -- the
// system that is left, the factor's panels, the forward-substituted right-hand side -- is read with L1-bypassing loads.
template <bool COH = false>
DEVI void dense_block_solve_body(const SolveArgs& a, int nsl, double* lds, const Handoff& ho, int t0 = 0, int outer_back = 0, int elim = 0, int code_pf = 0) {
  auto ldc = [](const double* p) { return COH ? load_sc1(p) : *p; };
  const long long t_entry = CAL_DEV_TIMING(a.debug != 0) ? __builtin_readcyclecounter() : 0;
  LmState* st = a.st;
  const int terminated = st->terminated;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lk = lane >> 4;
  const int m = a.m - t0, M1 = a.m + 1, n = a.n_s() + t0;
  const double* Sp = a.Spart + size_t(t0) * M1 + t0;      // entry (0, 0) of the system
  const int nb = (m + BP - 1) / BP;              // 32-column blocks
  const int mp = BP * nb;                        // padded size (identity beyond m)
  double* A = lds;                               // [128][DNL]
  double* gv = A + 128 * DNL;                    // [128] right-hand side row, forward-substituted in place (-> z)
  double* Daug = gv + 128;                       // [64][DLD] (rolling form: two compact channels, kDenseChan doubles)
  double* dinvm = Daug + kDenseChan;             // [128] diagonal of L⁻ᵀ
  double* yv = dinvm + 128;                      // [128] solution
  double* pend = yv + 128;                       // [128] Σ_{later blocks} Lᵀ y
  double* wv = pend + 128;                       // [32]
  double* bcast = wv + 32;                       // [128]
  double* dump = bcast + 128 + tid;              // [512]
  if (elim == 2) {
    // rolling owners: their channels (two compact ones in the panel buffer's place) and counters, cleared while the system's loads
    // are on their way -- the barrier that ends the load phase orders them
    for (int e = tid; e < kDenseChan; e += kDenseThreads) reinterpret_cast<unsigned long long*>(Daug)[e] = kElimSentinel;
    if (tid < 32) reinterpret_cast<int*>(bcast)[tid] = tid == 7 ? 2 : 0;      // (K_RESET = 7: both channels are clear)
    if (tid < 64) reinterpret_cast<long long*>(dump - tid)[tid] = 0;
  }
  // ---- load: lower triangle (K-slices summed), right-hand side = row m; identity beyond m ----
  {
    const size_t mm = size_t(M1) * M1;
    // (the right-hand side's loads go first, with the matrix's: requested behind the matrix's wait they were a second round trip)
    double rhs_acc = 0.0;
    if (tid < 128) {
      const int c = min(tid, m - 1);
      // (one request per slice there is: these two waves carry the longest share of the load phase's instructions)
      const double* const rp = Sp + size_t(m) * M1 + c;
      rhs_acc = 0.0 + ldc(rp);
      if (nsl > 1) rhs_acc += ldc(rp + mm);
      if (nsl > 2) {
#pragma unroll
        for (int k = 2; k < 8; ++k) rhs_acc += ldc(rp + size_t(min(k, nsl - 1)) * mm) * (k < nsl ? 1.0 : 0.0);
      }
    }
    // The lower triangle FOLDED into a rectangle of mp/2 rows of mp entries, dealt flat over the workgroup (sixteen entries per
    // thread at mp = 128): folded row f is row f with its diagonal (f + 1 entries) followed by the strict lower part of row
    // mp-1-f (mp-1-f entries); the diagonal of the rows mp/2.. is one more entry for threads 128.. . (A row per wave and
    // two columns per lane took twice the load instructions, half of them on clamped duplicates above the diagonal --
    // and the head of this launch is bound by the eight waves' load instructions, not by the round trip.)
    // (32-bit offsets, the second slice at a fixed distance, (row, column) kept packed between request and store: the
    //  instructions of this phase, two waves per SIMD, are what it takes -- not the round trip)
    const int half_rows = mp >> 1, n_fold = half_rows * mp;
    const int slice1 = nsl > 1 ? int(mm) : 0;       // (one slice: the same entry again, times zero)
    const double w1 = nsl > 1 ? 1.0 : 0.0;
    const bool diag_thread = tid >= 128 && tid < 128 + half_rows;
    const int rd = half_rows + (diag_thread ? tid - 128 : 0);
    double vdiag = 0.0;
    {
      const int rl = min(rd, m - 1), off = rl * M1 + rl;
      vdiag = (0.0 + ldc(Sp + off) * 1.0) + ldc(Sp + off + slice1) * w1;
    }
    // (the block count as a compile-time constant: thread tid's entries tid + 512 u then sit at folded row f0 + (16 / NB) u
    //  and a fixed position -- no division, and most of the index arithmetic folds)
    auto load_fold = [&](auto nb_tag, auto two_tag) {
      constexpr bool TWO = decltype(two_tag)::value;       // a second K-slice to add
      constexpr int NB = decltype(nb_tag)::value, MP = 32 * NB, NFOLD = MP / 2 * MP, NU = (NFOLD + kDenseThreads - 1) / kDenseThreads;
      double v[NU];
      int rc[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int e = min(tid + kDenseThreads * u, NFOLD - 1);
        const int f = (e >> 5) / NB, p = e - f * MP;
        const bool low = p <= f;
        const int r = low ? f : MP - 1 - f, c = low ? p : p - f - 1;
        rc[u] = r << 8 | c;
        const int rl = min(r, m - 1), cl = min(c, rl), off = rl * M1 + cl;
        v[u] = TWO ? (0.0 + ldc(Sp + off) * 1.0) + ldc(Sp + off + slice1) * w1 : ldc(Sp + off);
      }
      if (terminated) return;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int r = rc[u] >> 8, c = rc[u] & 255;
        if (tid + kDenseThreads * u < NFOLD) A[r * DNL + c] = (r < m) ? v[u] : (r == c ? 1.0 : 0.0);
      }
    };
    if (nsl > 1) {
      if (nb == 4) load_fold(std::integral_constant<int, 4>(), std::true_type());
      else if (nb == 3) load_fold(std::integral_constant<int, 3>(), std::true_type());
      else if (nb == 2) load_fold(std::integral_constant<int, 2>(), std::true_type());
      else load_fold(std::integral_constant<int, 1>(), std::true_type());
    } else {
      if (nb == 4) load_fold(std::integral_constant<int, 4>(), std::false_type());
      else if (nb == 3) load_fold(std::integral_constant<int, 3>(), std::false_type());
      else if (nb == 2) load_fold(std::integral_constant<int, 2>(), std::false_type());
      else load_fold(std::integral_constant<int, 1>(), std::false_type());
    }
    if (terminated) return;
    if (diag_thread) A[rd * DNL + rd] = rd < m ? vdiag : 1.0;
    if (tid < 128) {
      gv[tid] = tid < m ? rhs_acc : 0.0;
      pend[tid] = 0.0;
    }
  }
  if (terminated) return;
  if (nsl > 2) {
    // long trajectories: up to eight K-slices (reduced_schur_slices). Slices 2.. are added in a pass of their own over the
    // lower triangle, flat over the workgroup (the row-per-wave mapping above leaves most lanes on clamped duplicates)
    const size_t mm = size_t(M1) * M1;
    __syncthreads();
    for (int e0 = tid; e0 < m * m; e0 += kDenseThreads * 4) {
      double acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = min(e0 + kDenseThreads * u, m * m - 1), r = e / m, c = min(e - r * m, r);
        double sacc = 0.0;
#pragma unroll
        for (int k = 2; k < 8; ++k) sacc += Sp[size_t(min(k, nsl - 1)) * mm + size_t(r) * M1 + c] * (k < nsl ? 1.0 : 0.0);
        acc[u] = sacc;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + kDenseThreads * u, r = e / m, c = e - r * m;
        if (e < m * m && c <= r) A[r * DNL + c] += acc[u];
      }
    }
  }
  const bool dbg = CAL_DEV_TIMING(a.debug == 1 && (tid == 0 || tid == 64 * 5));
  long long tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = dbg ? __builtin_readcyclecounter() : 0;
#define DTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  asm volatile("" :: "v"(code_pf));      // (prefetch_code: the kernel's code, asked for at its first instructions)
  __syncthreads();
  DTICK(0)
  if (dbg) printf("dense_block_solve wave %d: %lld clocks from the kernel's first instruction to the loaded system\n", wave, (long long)(tk - t_entry));
  double pmin = 1.0;
  // 16x16 tile (I, c) of the trailing matrix minus the contribution of block jb's two panels, all sixteen operands
  // read before the first MFMA
  auto trail_tile = [&](int I, int c, int jb) {
    const double* pa = A + (16 * I + l16) * DNL + 32 * jb + lk;
    const double* pb = A + (16 * c + l16) * DNL + 32 * jb + lk;
    double va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { va[u] = -pa[4 * u]; vb[u] = pb[4 * u]; }
    double* pd = A + (16 * I + lk) * DNL + 16 * c + l16;
    f64x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = pd[4 * r * DNL];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], vb[u], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) pd[4 * r * DNL] = acc[r];
  };
  // the tiles behind block jb that the NEXT diagonal block does not need ("rest"), dealt to waves 1..7 in two
  // batches that run beside the two panels of the next block's factorisation (look-ahead)
  auto rest_batch = [&](int jb, int batch) {
    const int t_first = 2 * (jb + 1), T = mp / 16 - t_first;
    const int ntile = T * (T + 1) / 2;
    // tile q of the row-major lower triangle: q = 0, 1, 2 are (0,0), (1,0), (1,1) -- the next diagonal block
    // (six waves: wave 4 shares the panel wave's SIMD and stays out of its way, as in the tree levels)
    if (wave == 4) return;
    const int widx = wave < 4 ? wave - 1 : wave - 2;
    const int per_wave = (ntile - 3 + 5) / 6, half = (per_wave + 1) / 2;
    const int k0 = batch == 0 ? 0 : half, k1 = batch == 0 ? half : per_wave;
    for (int k = k0; k < k1; ++k) {
      const int q = 3 + widx + 6 * k;
      if (q >= ntile) break;
      int I = 0, rem = q;
      while (rem > I) { rem -= I + 1; ++I; }
      trail_tile(t_first + I, t_first + rem, jb);
    }
  };
  // diagonal block (lower part, zeros above) + identity rows into the panel buffer
  auto stage_diag = [&](int jb) {
    const int c0 = BP * jb;
#pragma unroll
    for (int u = 0; u < BB / kDenseThreads; ++u) {
      const int e = tid + kDenseThreads * u;
      const int r = e >> 5, c = e & 31;
      const double v = A[(c0 + r) * DNL + c0 + min(c, r)];
      Daug[r * DLD + c] = c <= r ? v : 0.0;
      Daug[(BP + r) * DLD + c] = r == c ? 1.0 : 0.0;
    }
  };
  if (elim == 2) {
    // ================================================================================================================
    // Rolling owners (round 6). The barrier form below spends, per 32-column block, 8.1k clocks in the elimination (chief 6.5k,
    // followers 1.3k behind) and 1.5k between two barriers on the right-hand side and the trailing update of the next block's
    // columns -- 9.6k for a dependent chain of 6.5k. Here wave j OWNS block-row j (rows 32j..32j+31) for the whole solve:
    //   * it follows the elimination of every block l < j with its rows of that block (elim_follow_owner) and keeps its own diagonal
    //     block D_j in registers, in the chief's layout, subtracting Z(j,l) Z(j,l)ᵀ step by step: when block j-1's last pivot is
    //     through, the wave holds D_j complete and goes on as block j's chief (elim_chief_reg) -- no barrier, no pass over LDS;
    //   * what block l does to the rows it will follow the NEXT block with, Z(j,l) Z(l+1,l)ᵀ, accumulates beside it in registers
    //     (the other operand: the step's columns of the next chief's rows, read from where that wave stored them, behind its
    //     progress word), so the owner's input for block l+1 is ready a few hundred clocks after block l is;
    //   * wave 4 follows every block with the identity rows (L⁻ᵀ) and the right-hand side, and forms the right-hand side of the
    //     next block's rows itself; wave 5 clears the channels behind the followers and carries the right-hand side of the rows
    //     further down; waves 0 and 7 the one trailing block nobody owns in time (rows of block 3 x columns of block 2, from block 0).
    // Two compact channels by block parity; order by single-writer counters (a wave's LDS instructions execute in order).
    // ================================================================================================================
    constexpr bool CC = true;
    double* const chan = Daug;                                    // [2][kElimCompactDoubles]
    int* const ctr = reinterpret_cast<int*>(bcast);               // [32]
    long long* const ts = reinterpret_cast<long long*>(dump - tid);      // [64] (dev timing)
    enum { K_PROG = 1, K_DONE = 2 /* + owner */, K_W4 = 6, K_RESET = 7, K_GV = 9, K_ID0 = 10, K_BG32 = 12 /* + tile: 12..15 */ };
    // (channels and counters were cleared in front of the load phase's barrier: see the top of the function)
    const long long t_roll = CAL_DEV_TIMING(a.debug == 1) ? __builtin_readcyclecounter() : 0;
    auto stamp = [&](int i) { if (CAL_DEV_TIMING(a.debug == 1 && lane == 0)) ts[i] = __builtin_readcyclecounter() - t_roll; };
    auto ctr_set = [&](int idx, int v) {
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(ctr + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
    };
    auto ctr_wait = [&](auto cond) {
      for (;;) {
        const int cv = __hip_atomic_load(ctr + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cond([&](int i) { return __builtin_amdgcn_readlane(cv, i); })) break;
        __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
    };
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    if (uw < 4) {
      const int j = uw;
      if (j < nb) {
        f64x4 d00, d01, d11;
        elim_load_spine(A + (BP * j) * DNL + BP * j, DNL, lane, d00, d01, d11);
        f64x4 x0[2] = {zero4, zero4}, x1[2] = {zero4, zero4}, nx0[2] = {zero4, zero4}, nx1[2] = {zero4, zero4};
        if (j > 0) {
          elim_load_rows(A + (BP * j) * DNL, DNL, 1, lane, x0[0], x1[0]);
          elim_load_rows(A + (BP * j + 16) * DNL, DNL, 1, lane, x0[1], x1[1]);
        }
        if (j > 1) {
          // the rows it will follow block 1 with: nobody else touches them, so block 0's share accumulates ON them and the input of
          // block 1's follow stands in registers when block 0's last step is through (loading them then and adding the share put
          // the next chief's follower 0.85k clocks behind before its first step)
          elim_load_rows(A + (BP * j) * DNL + BP, DNL, 1, lane, nx0[0], nx1[0]);
          elim_load_rows(A + (BP * j + 16) * DNL + BP, DNL, 1, lane, nx0[1], nx1[1]);
        }
        for (int l = 0; l <= j; ++l) {
          const int c0 = BP * l;
          const ElimChannel chl = elim_channel(chan + (l & 1) * kElimCompactDoubles);
          if (l >= 2) ctr_wait([&](auto c) { return c(K_RESET) >= l + 1; });
          if (l == j) {
            stamp(2 * l);
            elim_chief_reg<2, false, CC>(d00, d01, d11, A + c0 * DNL + c0, DNL, chl, lane);
            stamp(2 * l + 1);
            if (l == 0) {
              // block 0's identity rows (L⁻ᵀ: only the backward sweep reads it), by the one wave that has nothing left to do -- its
              // own channel is still there. Beside the first chief on its SIMD they made its steps 40 % longer.
              double* const blk = A;
              const ElimTile t[2] = {{gv, 0, 0, blk, DNL, 1, 1, dinvm}, {gv, 0, 0, blk + 16 * DNL, DNL, 1, 2, dinvm + 16}};
              elim_follow<2, CC>(t, chl, lane);
              ctr_set(K_ID0, 1);
            }
          } else {
            const bool cross = j > l + 1, dwave = j == l + 1;
            double* const out0 = A + (BP * j) * DNL + c0;
            const double* const c0p = A + (BP * (l + 1)) * DNL + c0;
            stamp(8 + 8 * j + 2 * l);
            elim_follow_owner<CC>(x0, x1, out0, out0 + 16 * DNL, DNL, chl, lane, d00, d01, d11, cross, nx0, nx1, c0p, c0p + 16 * DNL, dwave, ctr + K_PROG, 8 * l);
            ctr_set(K_DONE + j, l + 1);
            stamp(8 + 8 * j + 2 * l + 1);
            if (cross) {
              // the rows it follows block l+1 with: what the matrix holds (block 0's share of rows 3 x columns 2 comes from wave 6) + this block's share
              if (j == 3 && l == 1) ctr_wait([&](auto c) { return c(K_BG32) >= 1 && c(K_BG32 + 1) >= 1 && c(K_BG32 + 2) >= 1 && c(K_BG32 + 3) >= 1; });
              f64x4 b0, b1;
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                if (l == 0) { x0[q] = nx0[q]; x1[q] = nx1[q]; }      // (accumulated on the rows themselves: see the top)
                else {
                  elim_load_rows(A + (BP * j + 16 * q) * DNL + BP * (l + 1), DNL, 1, lane, b0, b1);
                  x0[q] = b0 + nx0[q]; x1[q] = b1 + nx1[q];
                }
                nx0[q] = zero4; nx1[q] = zero4;
              }
            }
          }
        }
      }
    } else if (uw == 4) {
      // identity rows (L⁻ᵀ: strict upper triangle + dinvm) and the right-hand side (in place in gv) of every block
      for (int l = 0; l < nb; ++l) {
        const int c0 = BP * l;
        const ElimChannel chl = elim_channel(chan + (l & 1) * kElimCompactDoubles);
        if (l >= 1) {
          // the right-hand side of this block's rows: minus Z(l, l-1) z_{l-1} (the rows further down: wave 5, a block earlier)
          ctr_wait([&](auto c) { return c(K_DONE + l) >= l && c(K_GV) >= l - 1 && c(K_RESET) >= l + 1; });
          const int r = lane & 31, h = lane >> 5;
          const double* zr = A + (c0 + r) * DNL + (c0 - BP) + 16 * h;
          const double* zv = gv + (c0 - BP) + 16 * h;
          double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 0; c < 16; ++c) a4[c & 3] += zr[c] * zv[c];
          double sum = (a4[0] + a4[1]) + (a4[2] + a4[3]);
          sum += other_half(sum);
          if (h == 0) gv[c0 + r] -= sum;
        }
        double* const blk = A + c0 * DNL + c0;
        const ElimTile t[3] = {{gv, 0, 0, blk, DNL, 1, 1, dinvm + c0}, {gv, 0, 0, blk + 16 * DNL, DNL, 1, 2, dinvm + c0 + 16},
                               {gv + c0, 0, 1, gv + c0, 0, 1, 3, nullptr}};
        stamp(40 + 2 * l);
        if (l == 0) { const ElimTile t0[1] = {{gv, 0, 1, gv, 0, 1, 3, nullptr}}; elim_follow<1, CC>(t0, chl, lane); }      // (identity rows: wave 0)
        else elim_follow<3, CC>(t, chl, lane);
        ctr_set(K_W4, l + 1);
        stamp(40 + 2 * l + 1);
      }
    } else if (uw == 5) {
      // behind every block: its channel cleared for the block after the next; the right-hand side of the rows two blocks down and further
      for (int l = 0; l + 2 < nb; ++l) {      // (the last two blocks' channels are nobody's any more, and no rows lie two blocks below them)
        ctr_wait([&](auto c) {
          bool ok = c(K_W4) >= l + 1 && (l > 0 || c(K_ID0) >= 1);
          for (int i = l + 1; i < nb; ++i) ok = ok && c(K_DONE + i) >= l + 1;
          return ok;
        });
        {
          unsigned long long* cp = reinterpret_cast<unsigned long long*>(chan + (l & 1) * kElimCompactDoubles);
          for (int e = lane; e < kElimCompactDoubles; e += 64) cp[e] = kElimSentinel;
        }
        ctr_set(K_RESET, l + 3);
        const int c0 = BP * l, row0 = c0 + 2 * BP;
        if (row0 < mp) {
          const int p = min(row0 + lane, mp - 1);
          const double* zr = A + p * DNL + c0;
          double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 0; c < BP; ++c) a4[c & 3] += zr[c] * gv[c0 + c];
          if (row0 + lane < mp) gv[p] -= (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        ctr_set(K_GV, l + 1);
      }
    }
    if (nb == 4 && (uw == 0 || uw == 7)) {
      // rows of block 3 x columns of block 2 minus block 0's share (block 1's accumulates in the owner's registers): four tiles, 512
      // clocks of matrix pipe each, while block 1 is eliminated. Three on wave 0 (behind its identity rows: its SIMD only carries the
      // right-hand side's follower by then), one on wave 7. Not on wave 5 or 6: one tile on wave 5 made step 2 of block 1's chief,
      // on its SIMD, 1.33k clocks instead of 0.83k (per-step stamps, CALICO_KERNEL_TIMING=1), and wave 6 sits beside the next chief.
      ctr_wait([&](auto c) { return c(K_DONE + 2) >= 1 && c(K_DONE + 3) >= 1; });
      if (uw == 7) { trail_tile(7, 4, 0); ctr_set(K_BG32 + 2, 1); }
      else {
        trail_tile(6, 4, 0); ctr_set(K_BG32, 1);
        trail_tile(6, 5, 0); ctr_set(K_BG32 + 1, 1);
        trail_tile(7, 5, 0); ctr_set(K_BG32 + 3, 1);
      }
    }
    lds_barrier();
    if (CAL_DEV_TIMING(a.debug == 1 && tid == 0)) {
      printf("dense_block_solve (rolling owners): chiefs begin-end %lld-%lld %lld-%lld %lld-%lld %lld-%lld, all through at %lld clocks behind the loaded system\n",
             ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], ts[7], (long long)(__builtin_readcyclecounter() - t_roll));
      printf("  owner 1 follows: %lld-%lld | owner 2: %lld-%lld %lld-%lld | owner 3: %lld-%lld %lld-%lld %lld-%lld | wave 4: %lld-%lld %lld-%lld %lld-%lld %lld-%lld\n",
             ts[16], ts[17], ts[24], ts[25], ts[26], ts[27], ts[32], ts[33], ts[34], ts[35], ts[36], ts[37], ts[40], ts[41], ts[42], ts[43], ts[44], ts[45], ts[46], ts[47]);
    }
  } else if (elim) {
    const ElimChannel ech = elim_channel(Daug);       // (the panel buffer is not used on this path)
    elim_reset(ech, tid, kDenseThreads);
    const int nt_all = mp / 16;
    for (int jb = 0; jb < nb; ++jb) {
      const int c0 = BP * jb;
      const int t_first = 2 * (jb + 1), nT = nt_all - t_first;      // row tiles below the block
      lds_barrier();
      DTICK(1)
      double* const blk = A + c0 * DNL + c0;
      if (wave == 0) elim_chief<2>(blk, DNL, ech, lane);
      else if (wave == 1) {
        const ElimTile t[3] = {{gv, 0, 0, blk, DNL, 1, 1, dinvm + c0}, {gv, 0, 0, blk + 16 * DNL, DNL, 1, 2, dinvm + c0 + 16},
                               {gv + c0, 0, 1, wv, 0, 1, 3, nullptr}};
        elim_follow<3>(t, ech, lane);
      } else if (wave == 2 || wave == 3) {
        // the row tiles below, dealt in halves: wave 2 takes the first ceil(nT / 2), wave 3 the rest (at most three each)
        const int h0 = (nT + 1) / 2;
        const int first = wave == 2 ? 0 : h0, cnt = wave == 2 ? h0 : nT - h0;
        double* const x0 = A + (16 * (t_first + first)) * DNL + c0;
        if (cnt == 3) {
          const ElimTile t[3] = {{x0, DNL, 1, x0, DNL, 1, 0, nullptr}, {x0 + 16 * DNL, DNL, 1, x0 + 16 * DNL, DNL, 1, 0, nullptr},
                                 {x0 + 32 * DNL, DNL, 1, x0 + 32 * DNL, DNL, 1, 0, nullptr}};
          elim_follow<3>(t, ech, lane);
        } else if (cnt == 2) {
          const ElimTile t[2] = {{x0, DNL, 1, x0, DNL, 1, 0, nullptr}, {x0 + 16 * DNL, DNL, 1, x0 + 16 * DNL, DNL, 1, 0, nullptr}};
          elim_follow<2>(t, ech, lane);
        } else if (cnt == 1) {
          const ElimTile t[1] = {{x0, DNL, 1, x0, DNL, 1, 0, nullptr}};
          elim_follow<1>(t, ech, lane);
        }
      } else if (jb > 0) {
        // trailing tiles of block jb-1 outside the current block's two column tiles (those were updated before the
        // barrier): tiles (I, c), c >= 2 relative to t_first - 2 ... i.e. column tiles >= t_first, rows I >= c; waves 5..7
        // (wave 4 shares the chief's SIMD)
        if (wave >= 5) {
          const int Tn = nt_all - t_first;                  // row / column tiles behind the current block
          const int ntile = Tn * (Tn + 1) / 2;
          for (int q = wave - 5; q < ntile; q += 3) {
            int I = 0, rem = q;
            while (rem > I) { rem -= I + 1; ++I; }
            trail_tile(t_first + I, t_first + rem, jb - 1);
          }
        }
      }
      lds_barrier();
      DTICK(4)
      // ---- between two blocks: the right-hand side, the channel, and the trailing update of the next block's columns ----
      if (tid < BP) gv[c0 + tid] = wv[tid];
      if (tid >= 256 && tid < 256 + mp - c0 - BP) {
        const int prow = c0 + BP + (tid - 256);
        const double* zr = A + prow * DNL + c0;
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < BP; ++c) a4[c & 3] += zr[c] * wv[c];
        gv[prow] -= (a4[0] + a4[1]) + (a4[2] + a4[3]);
      }
      if (jb + 1 < nb) {
        elim_reset(ech, tid, kDenseThreads);
        // tiles (I, c) with c in {t_first, t_first + 1}, I >= c: nT + (nT - 1) of them, over the eight waves
        const int n_next = 2 * nT - 1;
        for (int q = wave; q < n_next; q += 8) {
          const int c = q < nT ? 0 : 1, I = q < nT ? q : q - nT + 1;
          trail_tile(t_first + I, t_first + c, jb);
        }
      }
      DTICK(6)
    }
  } else {
  stage_diag(0);
  for (int jb = 0; jb < nb; ++jb) {
    const int c0 = BP * jb;
    lds_barrier();
    DTICK(1)
    if (wave == 0) panel_factor<1, false, false>(Daug, DLD, dinvm, bcast, 0, 63, 16, lane, &pmin);
    else if (jb > 0) rest_batch(jb - 1, 0);
    lds_barrier();
    DTICK(2)
    if (wave < 3) update_tile(Daug, DLD, 63, 1 + wave, 1, 0, 1, lane, dump);
    lds_barrier();
    DTICK(3)
    if (wave == 0) panel_factor<1, false, false>(Daug, DLD, dinvm, bcast, 16, 63, 16, lane, &pmin);
    else if (jb > 0) rest_batch(jb - 1, 1);
    lds_barrier();
    DTICK(4)
    const double* Mj = Daug + BP * DLD;           // L⁻ᵀ of this block (upper triangular)
    // ---- rows below: Z = X·M on the matrix cores, one wave per 16-row tile (both column tiles, then written in
    //      place); wave 7: the right-hand side row, z = g·M ----
    const int t_first = 2 * (jb + 1), t_end = mp / 16;
    for (int t = t_first + wave; wave < 7 && t < t_end; t += 7) {
      const double* X = A + (16 * t + l16) * DNL + c0 + lk;
      double xv[8], m0[4], m1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = X[4 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u) m0[u] = Mj[(4 * u + lk) * DLD + l16];
#pragma unroll
      for (int u = 0; u < 8; ++u) m1[u] = Mj[(4 * u + lk) * DLD + 16 + l16];
      f64x4 z0 = {0.0, 0.0, 0.0, 0.0}, z1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; ++u) z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u], m0[u], z0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u], m1[u], z1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* dst = A + (16 * t + lk + 4 * r) * DNL + c0 + l16;
        dst[0] = z0[r]; dst[16] = z1[r];
      }
    }
    if (wave == 7) {      // z = g·M, M upper triangular: lane (h, c) = (lane >> 5, lane & 31) sums rows 16h .. 16h+15 of column c
      // in four interleaved partial sums (a single 32-term chain with its LDS reads in between took longer than the
      // tile jobs beside it); fixed trip count, masked (a lane-dependent loop serialises)
      const int c = lane & 31, h = lane >> 5;
      double g16[16], mv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { g16[u] = gv[c0 + 16 * h + u]; mv[u] = Mj[(16 * h + u) * DLD + c]; }
      double z4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 16; ++u) z4[u & 3] += (16 * h + u <= c ? g16[u] : 0.0) * mv[u];
      double zc = (z4[0] + z4[1]) + (z4[2] + z4[3]);
      zc += other_half(zc);
      if (h == 0) wv[c] = zc;
    }
    lds_barrier();
    DTICK(5)
    if (tid < BP) gv[c0 + tid] = wv[tid];
    // L (lower) back into the matrix, L⁻ᵀ (strict upper) beside it, its diagonal apart: the backward sweep reads them
#pragma unroll
    for (int u = 0; u < BB / kDenseThreads; ++u) {
      const int e = tid + kDenseThreads * u;
      const int r = e >> 5, c = e & 31;
      const double lv = Daug[r * DLD + c], mv = Mj[r * DLD + c];
      A[(c0 + r) * DNL + c0 + c] = c <= r ? lv : mv;
      if (r == c) dinvm[c0 + r] = mv;
    }
    // right-hand side of the rows below: g_p -= Z_p · z
    if (tid >= 256 && tid < 256 + mp - c0 - BP) {
      const int prow = c0 + BP + (tid - 256);
      const double* zr = A + prow * DNL + c0;
      double a4[4] = {0.0, 0.0, 0.0, 0.0};      // four interleaved partial sums: no 32-term dependent chain
#pragma unroll
      for (int c = 0; c < BP; ++c) a4[c & 3] += zr[c] * wv[c];
      gv[prow] -= (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    // the three tiles of the next diagonal block first (waves 0..2): the factorisation of block jb+1 only waits for
    // these; the rest of the trailing update runs beside its panels
    if (jb + 1 < nb && wave < 3) trail_tile(t_first + (wave > 0 ? 1 : 0), t_first + (wave > 1 ? 1 : 0), jb);
    lds_barrier();
    DTICK(6)
    if (jb + 1 < nb) stage_diag(jb + 1);
  }
  }
  lds_barrier();
  // ---- backward: y_j = L_jj⁻ᵀ (z_j - pend_j), pend_k += L_jkᵀ y_j for the blocks before ----
  // (two barriers per block: the thread that completes pend of a column of the NEXT block to be solved forms that block's
  //  right-hand side z - pend at once -- a stage and a barrier of its own before)
  if (tid < BP) wv[tid] = gv[BP * (nb - 1) + tid] - pend[BP * (nb - 1) + tid];
  for (int jb = nb - 1; jb >= 0; --jb) {
    const int c0 = BP * jb;
    lds_barrier();
    if (tid < 8 * BP) {                      // eight threads per row, four columns each; fixed-shape reduction
      const int r = tid >> 3, part = tid & 7;
      const double* mr = A + (c0 + r) * DNL + c0;
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 4 * part + u;
        const double mv = c == r ? dinvm[c0 + r] : mr[c];
        acc += (c >= r ? mv : 0.0) * wv[c];
      }
      acc = row8_sum(acc);
      if (part == 0) yv[c0 + r] = acc;
    }
    lds_barrier();
    if (tid < 4 * c0) {                      // four threads per earlier column, eight rows each
      const int col = tid >> 2, part = tid & 3;
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int r = 8 * part + u; acc += A[(c0 + r) * DNL + col] * yv[c0 + r]; }
      acc = row4_sum(acc);
      if (part == 0) {
        const double pc = pend[col] + acc;
        pend[col] = pc;
        if (col >= c0 - BP) wv[col - (c0 - BP)] = gv[col] - pc;
      }
    }
  }
  lds_barrier();      // (y of block 0 is read by other threads below)
  DTICK(7)
  if (dbg) printf("dense_block_solve wave %d cycles: load %lld | stage %lld  panel0 (rest tiles beside it) %lld  tile %lld  panel1 (rest tiles) %lld  Z %lld  file+rhs+next diagonal %lld | backward %lld\n",
                  wave, tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6], tph[7]);
#undef DTICK
  if (t0 > 0 && outer_back) {
    // ---- the panels the step kernels eliminated, from the last to the first: y_j = L_jj⁻ᵀ (z_j - Σ_{i below} L_ijᵀ y_i).
    // L (rows below the panel: read along the rows, sixteen row groups), L_jj⁻ᵀ (filed by reduced_block_step_mfma_kernel
    // in the strict upper triangle of its diagonal block) and the forward-substituted right-hand side (row a.m of L) come
    // from global memory; the matrix area of this kernel is free by now. ----
    double* ybig = A;                    // [a.m <= 1024] solution by unknown
    double* ps = ybig + 1024;            // [16][33] partial sums of the row groups
    double* w32 = ps + 16 * 33;          // [32]
    double* Mt = w32 + 32;               // [32][33] L_jj⁻ᵀ
    const double* Lg = a.Swork;
    const int mt = a.m, ns0 = a.n_s();
    if (tid < m) ybig[t0 + tid] = yv[tid];
    for (int jb = t0 / BP - 1; jb >= 0; --jb) {
      const int c0 = BP * jb, c = tid & 31, g = tid >> 5;
      lds_barrier();
      double part = 0.0;
      for (int i0 = c0 + BP + g; i0 < mt; i0 += 64) {
        double lv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) lv[u] = ldc(Lg + size_t(min(i0 + 16 * u, mt - 1)) * M1 + c0 + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) part += i0 + 16 * u < mt ? lv[u] * ybig[i0 + 16 * u] : 0.0;
      }
      ps[g * 33 + c] = part;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + kDenseThreads * u, r = e >> 5, c2 = e & 31;
        const double v = ldc(Lg + size_t(c0 + r) * M1 + c0 + c2);
        Mt[r * 33 + c2] = c2 > r ? v : (c2 == r ? 1.0 / v : 0.0);
      }
      const double zq = tid < BP ? ldc(Lg + size_t(mt) * M1 + c0 + tid) : 0.0;
      lds_barrier();
      if (tid < BP) {
        double pd = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) pd += ps[q * 33 + tid];
        w32[tid] = zq - pd;
      }
      lds_barrier();
      if (tid < 8 * BP) {
        const int r = tid >> 3, part8 = tid & 7;
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += Mt[r * 33 + 4 * part8 + u] * w32[4 * part8 + u];
        acc = row8_sum(acc);
        if (part8 == 0) { ybig[c0 + r] = acc; a.y[ns0 + c0 + r] = acc; }
      }
    }
  }
  if (ho.word) {      // the back-substitution workgroups of this launch are waiting for the solution
    if (tid < m) store_sc1(a.y + n + tid, yv[tid]);
    handoff_publish(ho);
  } else if (tid < m) a.y[n + tid] = yv[tid];
  if (wave == 0 && lane == 0 && !(pmin > 0.0)) st->chol_failed = 1;
}
__global__ __launch_bounds__(kDenseThreads) void dense_block_solve_kernel(SolveArgs a, int nsl, int t0, int outer_back, int elim) {
  extern __shared__ double lds[];
  dense_block_solve_body(a, nsl, lds, Handoff{nullptr, 0}, t0, outer_back, elim);
}
// The dense reduced solve (workgroup 0) and the first back-substitution launch behind it (the other workgroups) in ONE
// launch: the nodes request everything they need that the reduced solve does not produce -- L⁻ᵀ, Z^A, Z^B, border rows,
// gradient, damping, current values: a few hundred loads per workgroup, ~3 us behind a kernel boundary -- while the solve
// is still running, wait for its hand-off, and go on with the solution. Saves a kernel boundary and the load phase.
static_assert(kDenseThreads == kBackThreads, "one launch, one workgroup size");
template <int QM, int MODE, bool PRE>
__global__ __launch_bounds__(kDenseThreads) void dense_back_kernel(SolveArgs a, BcrArgs b, int nsl, int node0, int n_nodes, int q_max,
                                                                   const double* __restrict__ x, double* __restrict__ x_cand,
                                                                   const BlockDev* __restrict__ blocks, int n_blocks, BcrTopSeps ts,
                                                                   int* word, int seq, int elim) {
  extern __shared__ double lds[];
  __shared__ double sh[64];
  const Handoff ho = {word, seq};
  const long long t_db = CAL_DEV_TIMING(a.debug == 4) ? __builtin_readcyclecounter() : 0;
  const int code_pf = 0;
  if (blockIdx.x == 0) {
    dense_block_solve_body(a, nsl, lds, ho, 0, 0, elim, code_pf);
    if (CAL_DEV_TIMING(a.debug == 4 && threadIdx.x == 0)) printf("dense_back: the solve's workgroup lived %lld clocks (terminated %d)\n", (long long)(__builtin_readcyclecounter() - t_db), a.st->terminated);
    return;
  }
  bcr_back_body<QM, MODE, true, PRE>(a, b, int(blockIdx.x) - 1, node0, n_nodes, 1, q_max, x, x_cand, blocks, n_blocks, ts, lds, sh, ho,
                                     node0 == 0 ? q_max : 0);      // (the nodes of this launch are level 0's: launch_dense_back)
  if (CAL_DEV_TIMING(a.debug == 4 && threadIdx.x == 0)) printf("dense_back: workgroup %d lived %lld clocks (terminated %d)\n", int(blockIdx.x), (long long)(__builtin_readcyclecounter() - t_db), a.st->terminated);
  asm volatile("" :: "v"(code_pf));
}
size_t dense_block_solve_lds_bytes() { return size_t(128 * DNL + 128 + kDenseChan + 128 * 3 + 32 + 128 + kDenseThreads) * sizeof(double); }
hipError_t configure_dense_block_solve() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_block_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             int(dense_block_solve_lds_bytes()));
}
void launch_dense_block_solve(const SolveArgs& a, int ks, hipStream_t s, int t0, int outer_back) {
  hipLaunchKernelGGL(dense_block_solve_kernel, dim3(1), dim3(kDenseThreads), dense_block_solve_lds_bytes(), s, a, ks, t0, outer_back, dense_elim_mode());
}
// ---------------------------------------------------------------------------
// Large reduced systems (m + 1 > 128): one step of the blocked right-looking factorisation, 32 columns, over several
// workgroups -- the same building blocks as a tree node. Workgroup (I, K), I >= K, owns the 64x64 tile of the trailing
// matrix at rows t0 + 64·I, columns t0 + 64·K (t0 = first row under the panel). Each workgroup factors the pivot
// block A_jj itself (two in-wave panels + one MFMA tile update, L⁻ᵀ riding along as 32 identity rows: identical in
// every workgroup, no communication), forms Zᵀ = L⁻¹ Pᵀ for its two row blocks P_I, P_K of the panel on the matrix
// cores and subtracts Z_I Z_Kᵀ from its tile in 16x16 MFMA tiles. The panel rows are read along the rows (a wave reads
// two 256-byte row segments per instruction) and transposed through LDS; the tile itself travels through the MFMA
// accumulators. Replaces reduced_block_step_kernel's 64-row in-wave column Cholesky (14k clocks per step) and its
// one-row-per-lane loads (64 cache lines per instruction: 21k clocks for the first step, which also sums the K-slices
// of the Schur complement). The first tile column files the panel into the factor L (Swork): L_jj by workgroup 0, the
// rows below by the workgroups with K = 0; the right-hand side rides as row m.
// ---------------------------------------------------------------------------
constexpr int kStepThreads = 512;
constexpr int PTL = 65;       // row stride of the transposed panel blocks [32][64]
// FUSED (reduced_fused_kernel: all steps and the in-LDS solve in one launch, step j + 1 behind a fan-in of step j's
// workgroups): the trailing matrix and the factor's panels leave with write-through stores and are read with L1-bypassing
// loads; `bid` is the workgroup's number within its step. Returns without a word on a terminated solve -- the caller arrives
// at the fan-in either way.
template <bool FUSED>
DEVI void reduced_block_step_body(const SolveArgs& a, int j, int nsl, int bid, double* lds) {
  auto ldA = [](const double* p) { return FUSED ? load_sc1(p) : *p; };
  auto stA = [](double* p, double v) { if (FUSED) store_sc1(p, v); else *p = v; };
  LmState* st = a.st;
  if (st->terminated) return;
  double* const Daug = lds;                        // [64][DLD]: rows 0..31 A_jj -> L_jj, rows 32..63 identity -> L⁻ᵀ
  double* const PT = Daug + 64 * DLD;              // [2][32][PTL]: panel rows of block rows I and K, transposed
  double* const ZT = PT + 2 * BP * PTL;            // [2][32][PTL]: Zᵀ = L⁻¹ Pᵀ
  double* const dinv = ZT + 2 * BP * PTL;          // [80]
  double* const bcast = dinv + 80;                 // [128]
  double* const dump = bcast + 128 + threadIdx.x;  // [512]
  const int m1 = a.m + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, lk = lane >> 4;
  const int c0 = BP * j, t0 = c0 + BP;
  int I = 0, rem = bid;
  while (rem > I) { rem -= I + 1; ++I; }
  const int K = rem;
  const int rI = t0 + 64 * I, rK = t0 + 64 * K;
  double* A = a.Spart;
  const size_t msq = size_t(m1) * m1;
  double* L = a.Swork;
  // ---- requests: pivot block, the two row blocks of the panel (along the rows), this wave's two tiles of the trailing
  //      matrix (straight into the accumulator layout) ----
  double dv[2], pv[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {      // pivot block: the lower triangle is stored, the upper one mirrored
    const int e = tid + kStepThreads * u, r = e >> 5, c = e & 31;
    const size_t o = size_t(c0 + max(r, c)) * m1 + c0 + min(r, c);
    double v = ldA(A + o);
    for (int k = 1; k < nsl; ++k) v += ldA(A + size_t(k) * msq + o);
    dv[u] = v;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rb = h == 0 ? rI : rK;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + kStepThreads * u, r = e >> 5, c = e & 31;
      const int row = rb + r;
      const size_t o = size_t(min(row, m1 - 1)) * m1 + c0 + c;
      double v = ldA(A + o);
      for (int k = 1; k < nsl; ++k) v += ldA(A + size_t(k) * msq + o);
      pv[h][u] = row < m1 ? v : 0.0;
    }
  }
  // tiles (ti, tj) of the 64x64 tile: wave w takes (w >> 1, 2 (w & 1)) and (w >> 1, 2 (w & 1) + 1)
  f64x4 tacc[2];
  const int ti = wave >> 1;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tj = 2 * (wave & 1) + q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ur = rI + 16 * ti + lk + 4 * r, uc = rK + 16 * tj + l16;
      const size_t o = size_t(min(ur, m1 - 1)) * m1 + min(uc, m1 - 1);
      double v = ldA(A + o);
      for (int k = 1; k < nsl; ++k) v += ldA(A + size_t(k) * msq + o);
      tacc[q][r] = v;
    }
  }
  // ---- to LDS ----
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + kStepThreads * u, r = e >> 5, c = e & 31;
    Daug[r * DLD + c] = dv[u];
    Daug[(BP + r) * DLD + c] = r == c ? 1.0 : 0.0;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + kStepThreads * u, r = e >> 5, c = e & 31;
      PT[(h * BP + c) * PTL + r] = pv[h][u];
    }
  __syncthreads();
  // ---- A_jj = L Lᵀ and L⁻ᵀ ----
  double pmin = 1.0;
  if (wave == 0) panel_factor<1, false, false>(Daug, DLD, dinv, bcast, 0, 63, 16, lane, &pmin);
  lds_barrier();
  if (wave < 3) update_tile(Daug, DLD, 63, 1 + wave, 1, 0, 1, lane, dump);
  lds_barrier();
  if (wave == 0) panel_factor<1, false, false>(Daug, DLD, dinv, bcast, 16, 63, 16, lane, &pmin);
  lds_barrier();
  // ---- Zᵀ = L⁻¹ Pᵀ = Mᵀ Pᵀ, M = L⁻ᵀ in rows 32..63 (upper triangular: row tile it needs k < 16(it+1)) ----
  {
    const double* M = Daug + BP * DLD;
    // 16 jobs (h, it, jt): block row h, row tile it of Zᵀ (the panel's columns), column tile jt (the block's rows)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int job = wave + 8 * q, h = job >> 3, it = (job >> 2) & 1, jt = job & 3;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      acc = atb_tile<false>(M, DLD, 16 * it, PT + h * BP * PTL, PTL, 16 * jt, 0, 16 * (it + 1), acc, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) ZT[(h * BP + 16 * it + lk + 4 * r) * PTL + 16 * jt + l16] = acc[r];
    }
  }
  lds_barrier();
  // ---- file the panel: L_jj (workgroup 0), the rows below (first tile column) ----
  if (bid == 0) {      // (L⁻ᵀ in the strict upper triangle beside L: the backward sweep multiplies by it, its diagonal is 1 / L_rr)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + kStepThreads * u, r = e >> 5, c = e & 31;
      stA(L + size_t(c0 + r) * m1 + c0 + c, c <= r ? Daug[r * DLD + c] : Daug[(BP + r) * DLD + c]);
    }
  }
  if (K == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + kStepThreads * u, r = e >> 5, c = e & 31;
      if (rI + r < m1) stA(L + size_t(rI + r) * m1 + c0 + c, ZT[c * PTL + r]);
    }
  }
  // ---- tile update: A_IK -= Z_I Z_Kᵀ ----
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tj = 2 * (wave & 1) + q;
    tacc[q] = atb_tile<true>(ZT, PTL, 16 * ti, ZT + BP * PTL, PTL, 16 * tj, 0, BP, tacc[q], lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ur = rI + 16 * ti + lk + 4 * r, uc = rK + 16 * tj + l16;
      if (ur < m1 && uc <= ur) stA(A + size_t(ur) * m1 + uc, tacc[q][r]);
    }
  }
}
__global__ __launch_bounds__(kStepThreads) void reduced_block_step_mfma_kernel(SolveArgs a, int j, int nsl) {
  extern __shared__ double lds[];
  reduced_block_step_body<false>(a, j, nsl, int(blockIdx.x), lds);
}
// Workgroups of step j of the blocked factorisation (T(T+1)/2 tiles of 64x64 below the panel, at least one)
__host__ __device__ inline int reduced_step_workgroups(int m1, int j) {
  const int rows = m1 - BP * (j + 1), T = rows > 0 ? (rows + 63) / 64 : 0;
  return T > 0 ? T * (T + 1) / 2 : 1;
}
// The whole blocked factorisation of a reduced system of more than 128 columns in ONE launch: the workgroups of step j + 1
// wait for those of step j at a fan-in word (words[j]; every workgroup arrives, terminated solve or not), the last
// workgroup is the in-LDS solver of what the steps leave (dense_block_solve_body with the blocked backward sweep behind
// it) and waits for the last step. All of them are resident at once (launch_reduced_fused checks the count), so nobody
// waits for a workgroup that has no CU. The solver's workgroup clears the words behind its wait: every other waiter has
// passed its own by then (it has arrived at a later word), and the next launch finds zeros.
static_assert(kStepThreads == kDenseThreads, "one launch, one workgroup size");
__global__ __launch_bounds__(kStepThreads) void reduced_fused_kernel(SolveArgs a, int nsteps, int nsl, int* words, int elim) {
  extern __shared__ double lds[];
  const int m1 = a.m + 1;
  int bid = blockIdx.x, j = 0;
  for (; j < nsteps; ++j) { const int nw = reduced_step_workgroups(m1, j); if (bid < nw) break; bid -= nw; }
  if (j < nsteps) {
    if (j > 0) fanin_wait(words + j - 1, reduced_step_workgroups(m1, j - 1));
    reduced_block_step_body<true>(a, j, j == 0 ? nsl : 1, bid, lds);
    fanin_arrive(words + j);
    return;
  }
  fanin_wait(words + nsteps - 1, reduced_step_workgroups(m1, nsteps - 1));
  if (int(threadIdx.x) < nsteps) __hip_atomic_store(words + threadIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dense_block_solve_body<true>(a, 1, lds, Handoff{nullptr, 0}, BP * nsteps, 1, elim);
}
size_t reduced_block_step_lds_bytes() { return size_t(64 * DLD + 4 * BP * PTL + 80 + 128 + kStepThreads) * sizeof(double); }
hipError_t configure_reduced_block_step() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&reduced_block_step_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             int(reduced_block_step_lds_bytes()));
}
size_t reduced_fused_lds_bytes() { return std::max(reduced_block_step_lds_bytes(), dense_block_solve_lds_bytes()); }
hipError_t configure_reduced_fused() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&reduced_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(reduced_fused_lds_bytes()));
}
// false: too many workgroups to count on all of them being resident together (one per CU with the solver's LDS footprint)
bool launch_reduced_fused(const SolveArgs& a, int nsteps, int nsl, int* words, hipStream_t s) {
  int n_wg = 1;
  for (int j = 0; j < nsteps; ++j) n_wg += reduced_step_workgroups(a.m + 1, j);
  if (nsteps < 1 || nsteps > 8 || n_wg > 128) return false;
  hipLaunchKernelGGL(reduced_fused_kernel, dim3(n_wg), dim3(kStepThreads), reduced_fused_lds_bytes(), s, a, nsteps, nsl, words, dense_elim_mode());
  return true;
}
void launch_reduced_block_step(const SolveArgs& a, int j, int nsl, int n_wg, hipStream_t s) {
  hipLaunchKernelGGL(reduced_block_step_mfma_kernel, dim3(n_wg), dim3(kStepThreads), reduced_block_step_lds_bytes(), s, a, j, nsl);
}
size_t bcr_back_lds_bytes(int q_max, int m1p);
// Does the Schur complement ride in the last level's launch? Trees of at least two levels whose last level has one or two
// single-superblock nodes (it always has, by construction of the plan); CALICO_FUSE_SCHUR=0: a launch of its own (A/B).
bool schur_rides_in_last_level(int n_levels, int n_last_nodes, int root) {
  const char* e = std::getenv("CALICO_FUSE_SCHUR");
  return (!e || std::atoi(e) != 0) && n_levels >= 2 && n_last_nodes >= 1 && n_last_nodes <= 2 && root >= 0;
}
// Can the first back-substitution launch ride in the dense solve's launch? Only the shapes the in-LDS solve takes, no
// border-row sweep workgroups (those would sit on every CU with the dense solve's LDS footprint), chains of at most four.
bool dense_back_fusable(const SolveArgs& a, int ks, int q_max, bool border_rows) {
  const char* fe = std::getenv("CALICO_FUSE_BACK");       // (read per solve: an A/B switch, and what the tests toggle)
  const bool on = !fe || std::atoi(fe) != 0;
  static const bool use_block = [] { const char* e = std::getenv("CALICO_DENSE"); return !(e && std::string(e) == "panel"); }();
  return on && use_block && a.m + 1 <= 128 && a.m >= 1 && ks <= 2 && q_max <= 4 && !border_rows;
}
// PRE: the nodes form their solution as an affine map of the reduced solve's output while they wait (back_node_pre);
// needs mc + 33 <= 128 columns (one 16-column tile per wave) and a reduced solve long enough to hide the recursion behind:
// it takes ~50k clocks (requests 15k, staging 15k, recursion 20k; dev-timing dump), a reduced solve 11k + 16k per
// 32-column block -- four blocks (m + 1 > 96: configs[3]) cover it, two (configs[1]: 9450 with, 10200 it/s without) do
// not. CALICO_BACK_PRE=0 / 1: never / whenever the columns fit (A/B switch).
static bool dense_back_pre(const SolveArgs& a) {
  const char* e = std::getenv("CALICO_BACK_PRE");
  if (a.mc + BP + 1 > 128) return false;
  if (e) return std::atoi(e) != 0;
  return a.m + 1 > 96;
}
static size_t dense_back_lds(int q_max, int m1p) {
  const int qm = q_max <= 1 ? 1 : (q_max <= 2 ? 2 : 4);
  return std::max(std::max(dense_block_solve_lds_bytes(), bcr_back_lds_bytes(q_max, m1p)), bcr_back_pre_lds_doubles(qm) * sizeof(double));
}
size_t dense_back_lds_bytes(int q_max, int m1p) { return dense_back_lds(q_max, m1p); }
hipError_t configure_dense_back_bytes(size_t lds) {
  for (const void* f : {reinterpret_cast<const void*>(&dense_back_kernel<1, 1, false>), reinterpret_cast<const void*>(&dense_back_kernel<2, 1, false>),
                        reinterpret_cast<const void*>(&dense_back_kernel<4, 1, false>), reinterpret_cast<const void*>(&dense_back_kernel<1, 2, false>),
                        reinterpret_cast<const void*>(&dense_back_kernel<2, 2, false>), reinterpret_cast<const void*>(&dense_back_kernel<4, 2, false>),
                        reinterpret_cast<const void*>(&dense_back_kernel<1, 1, true>), reinterpret_cast<const void*>(&dense_back_kernel<2, 1, true>),
                        reinterpret_cast<const void*>(&dense_back_kernel<4, 1, true>), reinterpret_cast<const void*>(&dense_back_kernel<1, 2, true>),
                        reinterpret_cast<const void*>(&dense_back_kernel<2, 2, true>), reinterpret_cast<const void*>(&dense_back_kernel<4, 2, true>)}) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
void launch_dense_back(const SolveArgs& a, const BcrArgs& b, int ks, int node0, int n_nodes, int q_max, const double* x, double* x_cand,
                       const BlockDev* blocks, int n_blocks, const BcrTopSeps& ts, int* word, int seq, hipStream_t s) {
  const size_t lds = dense_back_lds(q_max, b.m1p);
  const dim3 grid(1 + n_nodes + 1), block(kDenseThreads);       // dense solve, the nodes, the calibration / root update
  const int elim = dense_elim_mode();
#define LAUNCH_DB(QM, SD, PR) hipLaunchKernelGGL(HIP_KERNEL_NAME(dense_back_kernel<QM, SD, PR>), grid, block, lds, s, a, b, ks, node0, n_nodes, q_max, x, x_cand, blocks, n_blocks, ts, word, seq, elim)
  if (dense_back_pre(a)) {
    if (ts.n > 0) { if (q_max <= 1) LAUNCH_DB(1, 2, true); else if (q_max <= 2) LAUNCH_DB(2, 2, true); else LAUNCH_DB(4, 2, true); }
    else { if (q_max <= 1) LAUNCH_DB(1, 1, true); else if (q_max <= 2) LAUNCH_DB(2, 1, true); else LAUNCH_DB(4, 1, true); }
  } else {
    if (ts.n > 0) { if (q_max <= 1) LAUNCH_DB(1, 2, false); else if (q_max <= 2) LAUNCH_DB(2, 2, false); else LAUNCH_DB(4, 2, false); }
    else { if (q_max <= 1) LAUNCH_DB(1, 1, false); else if (q_max <= 2) LAUNCH_DB(2, 1, false); else LAUNCH_DB(4, 1, false); }
  }
#undef LAUNCH_DB
}

// ---- launch helpers ---------------------------------------------------------
// CALICO_ELIM=panel: the block factorisation of rounds 1-3 (two in-wave panels + tile update + Z phase); read per solve (A/B switch)
bool block_elim_enabled() { const char* e = std::getenv("CALICO_ELIM"); return !(e && std::string(e) == "panel"); }
// The dense reduced solve with rolling owners (dense_block_solve_body, elim == 2), the default; CALICO_DENSE_ROLL=0: the barrier
// form (A/B switch, read per solve)
static int dense_elim_mode() {
  if (!block_elim_enabled()) return 0;
  const char* e = std::getenv("CALICO_DENSE_ROLL");
  return (!e || std::atoi(e) != 0) ? 2 : 1;
}
// CALICO_LOOKAHEAD=1: the tree levels' steps with look-ahead (bcr_level_kernel<.., LA>); read per solve (A/B switch). OFF by
// default: bit-identical, but measured 1.2 us SLOWER per level-0 launch at configs[3] (26.2 against 25.0 us, same box,
// profiles/r05_lookahead_ab.txt) -- the chief does start ~2.5k clocks earlier per step, but a step is then bounded by the
// loader waves' commit of the next block (they lose the Schur phase as load time) and by the two barriers' own latency.
static bool level_lookahead_enabled() { const char* e = std::getenv("CALICO_LOOKAHEAD"); return e && std::atoi(e) != 0; }
// Level 0's chains with the rolling chief (bcr_level_kernel<true, true, false, true>: no workgroup barrier between the blocks of
// a chain), the default; CALICO_ROLL=0: the barrier form (A/B switch, read per solve)
static bool level_roll_enabled() { const char* e = std::getenv("CALICO_ROLL"); return !e || std::atoi(e) != 0; }
// the same form on the levels above level 0 (single-block chains): CALICO_ROLL_UPPER=1
// (NOT the default: the chains of an upper level end 2-3k clocks earlier with it, the launch does not -- it ends with the Schur
//  complement's riders behind the fan-in --, and the iteration rate is 0.7-1.5 % lower at every shape measured.)
static bool level_roll_upper_enabled() { const char* e = std::getenv("CALICO_ROLL_UPPER"); return e && std::atoi(e) != 0; }
size_t bcr_level_lds_bytes() { return size_t(2 * 64 * DLD + 4 * BP * XLD + 80 + 128 + kLevelThreads + kElimBufDoubles) * sizeof(double); }      // (4: X twice, Z twice with the look-ahead)
size_t bcr_back_lds_bytes(int q_max, int m1p) {
  return (size_t(2) * q_max * BP * DLD + kBcrMaxChain * BP + 3 * BP + m1p + size_t(q_max) * BP + 4 * BP) * sizeof(double);
}
hipError_t configure_bcr_kernels(int q_max, int m1p) {
  hipError_t e = upload_roll_table();      // (per device: a __device__ symbol lives on each of them)
  if (e != hipSuccess) return e;
  for (const void* f : {reinterpret_cast<const void*>(&bcr_level_kernel<true, true>), reinterpret_cast<const void*>(&bcr_level_kernel<false, true>),
                        reinterpret_cast<const void*>(&bcr_level_kernel<true, true, true>), reinterpret_cast<const void*>(&bcr_level_kernel<false, true, true>),
                        reinterpret_cast<const void*>(&bcr_level_kernel<true, false>), reinterpret_cast<const void*>(&bcr_level_kernel<false, false>),
                        reinterpret_cast<const void*>(&bcr_level_kernel<true, true, false, true>), reinterpret_cast<const void*>(&bcr_level_kernel<false, true, false, true>)}) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(bcr_level_lds_bytes()));
    if (e != hipSuccess) return e;
  }
  for (const void* f : {reinterpret_cast<const void*>(&bcr_back_kernel<1, 0>), reinterpret_cast<const void*>(&bcr_back_kernel<2, 0>),
                        reinterpret_cast<const void*>(&bcr_back_kernel<4, 0>), reinterpret_cast<const void*>(&bcr_back_kernel<8, 0>),
                        reinterpret_cast<const void*>(&bcr_back_kernel<1, 1>), reinterpret_cast<const void*>(&bcr_back_kernel<2, 1>),
                        reinterpret_cast<const void*>(&bcr_back_kernel<4, 1>), reinterpret_cast<const void*>(&bcr_back_kernel<8, 1>),
                        reinterpret_cast<const void*>(&bcr_back_kernel<1, 2>), reinterpret_cast<const void*>(&bcr_back_kernel<2, 2>),
                        reinterpret_cast<const void*>(&bcr_back_kernel<4, 2>)}) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(bcr_back_lds_bytes(q_max, m1p)));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// `schur_ks` > 0 (the LAST level of a tree of at least two): the Schur complement's tiles and the root's rows ride behind
// this level's workgroups and take its results over the fan-in word; level 0 (`fan_word` given) resets the word.
void launch_bcr_level(const SolveArgs& a, const BcrArgs& b, int node0, int n_nodes, int level, int keep0, int n_keep, const LmOptionsDev& o,
                      const double* x, const BlockDev* blocks, int n_blocks, int with_post_eval, IterLog* log, int log_cap, int jacobi,
                      hipStream_t s, int schur_ks, int* fan_word, const BcrInlineNodes& inl, int q_max) {
  const int nfs = (a.mc + 1 + kBcrFS - 1) / kBcrFS;
  int n_apply = n_keep > 0 ? std::min(64, std::max(1, n_keep * 4)) : 0;
  const int main_span = 8 * ((n_nodes + 7) / 8) * (1 + nfs);      // (node, role) workgroups laid out by XCD: see the kernel
  // One workgroup of this kernel fills a CU (its LDS), and the part has 256 of them: where the chains leave room, the separators'
  // workgroups (and level 0's bookkeeping workgroup, the last of the grid) are kept inside that room -- a workgroup that is only
  // dispatched when another has ended starts ~15k clocks late, and the bookkeeping one then ends the launch (440 control points:
  // 224 + 64 + 1 workgroups, level 0 21.5 us with the chains done at 13 us).
  {
    constexpr int kCUs = 256;
    const int room = kCUs - main_span - (level == 0 && with_post_eval ? 1 : 0);
    if (room >= 8 && n_apply > room) n_apply = room;
  }
  const bool elim = block_elim_enabled(), la = level_lookahead_enabled();
  if (level == 0) {
    const bool roll = elim && !la && a.k >= 1 && a.k <= 6 && level_roll_enabled();
    hipLaunchKernelGGL((roll ? bcr_level_kernel<true, true, false, true> : elim ? (la ? bcr_level_kernel<true, true, true> : bcr_level_kernel<true, true>) : bcr_level_kernel<true, false>), dim3(main_span + n_apply + (with_post_eval ? 1 : 0)), dim3(kLevelThreads),
                       bcr_level_lds_bytes(), s, a, b, node0, n_nodes, nfs, level, keep0, n_keep, o, with_post_eval, x, blocks, n_blocks,
                       log, log_cap, jacobi, 0, 0, 1, fan_word, 0, inl);
  } else {
    const int nt = (a.mc + 1 + 15) / 16;
    const int br = a.m - a.mc;
    // (the root's rows: one entry per thread, so that the riders' tail behind the fan-in is a single round trip)
    const int n_schur_wg = schur_ks > 0 ? nt * (nt + 1) / 2 * schur_ks : 0;
    const int n_root_wg = schur_ks > 0 ? std::max(1, (br * (a.mc + 1 + br) + kLevelThreads - 1) / kLevelThreads) : 0;
    const int n_prod = n_nodes * (1 + nfs) + n_apply;        // the workgroups of this level that really exist
    const bool roll = elim && !la && q_max == 1 && level_roll_enabled() && level_roll_upper_enabled();      // (single-block chains: see the kernel)
    hipLaunchKernelGGL((roll ? bcr_level_kernel<false, true, false, true> : elim ? (la ? bcr_level_kernel<false, true, true> : bcr_level_kernel<false, true>) : bcr_level_kernel<false, false>), dim3(main_span + n_apply + n_schur_wg + n_root_wg), dim3(kLevelThreads), bcr_level_lds_bytes(), s, a, b,
                       node0, n_nodes, nfs, level, keep0, n_keep, o, 0, x, blocks, n_blocks, log, log_cap, jacobi, n_schur_wg, n_root_wg,
                       std::max(1, schur_ks), schur_ks > 0 ? fan_word : nullptr, n_prod, inl);
  }
}
void launch_bcr_schur(const SolveArgs& a, const BcrArgs& b, int ks, const LmOptionsDev& o, hipStream_t s) {
  const int nt = (a.mc + 1 + 15) / 16;
  const int n_tile_wg = nt * (nt + 1) / 2 * ks;
  hipLaunchKernelGGL(bcr_schur_kernel, dim3(n_tile_wg + (b.root >= 0 ? 4 : 0)), dim3(256), 0, s, a, b, ks, n_tile_wg, o);
}
void launch_bcr_back(const SolveArgs& a, const BcrArgs& b, int node0, int n_nodes, bool top, bool extras, bool border_rows, int q_max,
                     const double* x, double* x_cand, const BlockDev* blocks, int n_blocks, const BcrTopSeps& ts, hipStream_t s) {
  const int n_mv = border_rows ? (b.N * BP + kBackThreads / 64 - 1) / (kBackThreads / 64) : 0;   // (nobody below reads b.zb: no sweep)
  const dim3 grid(n_nodes + (extras ? 1 + n_mv : 0)), block(kBackThreads);
  const size_t lds = bcr_back_lds_bytes(q_max, b.m1p);
#define LAUNCH_BACK(QM) hipLaunchKernelGGL(HIP_KERNEL_NAME(bcr_back_kernel<QM, SD>), grid, block, lds, s, a, b, node0, n_nodes, top ? 1 : 0, extras ? 1 : 0, q_max, x, x_cand, blocks, n_blocks, ts)
  if (ts.n > 0) {      // (the caller keeps chains longer than four out of this mode: the registers do not hold both)
    constexpr int SD = 2;
    if (q_max <= 1) LAUNCH_BACK(1); else if (q_max <= 2) LAUNCH_BACK(2); else LAUNCH_BACK(4);
  } else if (top) {
    constexpr int SD = 1;
    if (q_max <= 1) LAUNCH_BACK(1); else if (q_max <= 2) LAUNCH_BACK(2); else if (q_max <= 4) LAUNCH_BACK(4); else LAUNCH_BACK(8);
  } else {
    constexpr int SD = 0;
    if (q_max <= 1) LAUNCH_BACK(1); else if (q_max <= 2) LAUNCH_BACK(2); else if (q_max <= 4) LAUNCH_BACK(4); else LAUNCH_BACK(8);
  }
#undef LAUNCH_BACK
}

}  // namespace cal
