// block_elim.hpp — elimination of one 32x32 diagonal block on the matrix cores, spread over the SIMDs of a CU (round 4).
//
// What the tree levels, the dense reduced solve and the blocked reduced factorisation all do per step is
//     D = L Lᵀ,   M = L⁻ᵀ,   Z = L⁻¹ X   (X: the columns that couple to the block; or Z = X L⁻ᵀ for the rows below it),
// and rounds 1-3 did it as: two in-wave 16-column panels (one lane per row of [D; I], 31 instructions per column on ONE
// wave) | barrier | MFMA tile update | barrier | panel | barrier | Z = MᵀX on the matrix cores | barrier: ~11.5k clocks
// per block inside the kernels, the panels alone ~9.3k.
//
// Here the block is eliminated four columns per step, eight steps, with everything in MFMA accumulator tiles held
// TRANSPOSED: tile (J, T) of a row tile T (sixteen rows of [D; I; Xᵀ]) and column tile J (sixteen columns of the block),
//     acc[r] of lane (l16 = lane & 15, lk = lane >> 4)  =  -Aug[16T + l16][16J + lk + 4r]        (negated).
// In this form
//   * register u of a tile IS the operand "sixteen rows x the four columns of step u" of v_mfma_f64_16x16x4_f64 (lane
//     (l16, lk) holds entry (l16, k = lk)) -- as A operand and as B operand alike;
//   * the step's four columns of the factor for row tile T, L_T = Aug_T,step · L44⁻ᵀ, are ONE MFMA with -L44⁻¹ (the
//     inverse of the 4x4 pivot block's Cholesky factor, rows 0..3 of a 16x4 A operand) against register u; result
//     register 0 of lane (l16, lk) is L[16T + l16][step column lk] -- the operand layout again -- so the trailing update
//     tile(J', T) += L_J' L_Tᵀ takes both operands straight from result registers. No LDS, no lane movement;
//   * the only cross-lane traffic is the 4x4 pivot block (ten v_readlane pairs out of register u of the diagonal tile),
//     factored redundantly by every lane: 4 x [v_rsq_f64 + one Newton step] on the chain; the lane's entry of -L44⁻¹ is
//     the forward substitution of its own unit vector e_lk, selected by l16.
// One wave cannot go faster than its chain (eight steps of ~75 VALU instructions + two dependent MFMAs) OR than its
// flops (FP64 MFMA: 64 clocks per 16x16x4 on this part -- the whole augmented block is ~50 of them), so the work is split:
//   * the CHIEF wave owns the spine (the block itself: tiles (0,0), (0,1), (1,1)) and nothing else -- five MFMAs per
//     step -- and publishes per step, write-once in LDS: w (the pivot operand), l0 / l1 (the step's columns of L for
//     rows 0..15 / 16..31), then a progress word;
//   * FOLLOWER waves on the other SIMDs own the remaining row tiles (identity rows -> M, Xᵀ rows -> Zᵀ, rows below -> Z):
//     per step and row tile one panel MFMA (whose result is final: four rows of Z, written out at once) and up to two
//     trailing updates. They run behind the chief by a fraction of a step; nobody waits for them until the end.
// Z therefore comes out of the factorisation itself: the separate "Z = MᵀX" phase and its barrier are gone, and M is
// only formed where somebody files it.
// Rows of a diagonal tile above the current pivot block carry rounding residue of entries that are zero in exact
// arithmetic (A - L Lᵀ over the eliminated part); their products only land in entries nobody reads again.
// A bad pivot is not patched: NaN propagates and is caught by the update stage.
#pragma once
#include "solve_dev.hpp"

namespace cal {

constexpr int kElimSlot = 64;                          // doubles per published vector (one per lane)
constexpr int kElimStep = 3 * kElimSlot;               // w, l0, l1
constexpr int kElimBufDoubles = 8 * kElimStep;         // write-once per factorisation
// No flag and no wait on the chief's side: the channel is filled with a sentinel (a NaN no arithmetic produces) before a
// factorisation, the chief stores w, l0, l1 of a step in that order, and a follower requests l1, w, l0 in THAT order and
// looks at its own lane's l1: LDS executes instructions in order, so once every lane has seen its l1 the reads behind it
// have seen w and l0. (A progress word needed s_waitcnt lgkmcnt(0) in front of its store: the chief's stores queue behind
// the followers' polling reads, and the wait cost it ~100 clocks per step inside the kernels.)
constexpr unsigned long long kElimSentinel = 0x7FF8E11AE11AE11Aull;

// Compact layout (CC, round 6: the dense reduced solve keeps two channels where it had room for one): only the vectors somebody
// reads -- w of the eight steps, l0 of steps 0..2, l1 of steps 0..6: 18 instead of 24.
constexpr int kElimCompactDoubles = 18 * kElimSlot;
template <bool CC> DEVI constexpr int elim_slot_w(int s) { return (CC ? s : 3 * s) * kElimSlot; }
template <bool CC> DEVI constexpr int elim_slot_l0(int s) { return (CC ? 8 + s : 3 * s + 1) * kElimSlot; }
template <bool CC> DEVI constexpr int elim_slot_l1(int s) { return (CC ? 11 + s : 3 * s + 2) * kElimSlot; }
struct ElimChannel { double* buf; };      // LDS [8][3][64] (compact: [18][64])
DEVI ElimChannel elim_channel(double* lds /* kElimBufDoubles */) { return ElimChannel{lds}; }
// all threads of the workgroup (any time after the followers of the last factorisation are through; a barrier follows)
DEVI void elim_reset(const ElimChannel& ch, int tid, int nthreads) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(ch.buf);
  for (int e = tid; e < kElimBufDoubles; e += nthreads) p[e] = kElimSentinel;
}
DEVI void elim_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");       // (the order of the stores of a step is part of the protocol)
}

// The 4x4 pivot block of a step, factored by every lane, in seven short stages: the matrix pipe is in-order and a queued
// MFMA blocks the wave's issue, so the chief's MFMAs that are off the chain are placed BETWEEN these stages, with
// scheduling barriers -- left to the compiler they all land in front of the next step's critical panel MFMA.
struct PivotChain {
  double d;                 // register u of the (negated) diagonal tile
  int b, l16;               // lane of the pivot block's entry (0, 0); the lane's l16
  double e0, e1, e2, e3;    // -e_lk
  double a10, a11, a20, a21, a22, a30, a31, a32, a33;
  double r0, r1, r2, r3, l10, l20, l30, l21, l31, l32, x3a, x0, x1, x2;
  DEVI void s0() { const double a00 = -readlane_f64(d, b); r0 = rsqrt_nr(a00); a10 = -readlane_f64(d, 16 + b); a11 = -readlane_f64(d, 17 + b); }
  DEVI void s1() { l10 = a10 * r0; r1 = rsqrt_nr(__builtin_fma(-l10, l10, a11)); a20 = -readlane_f64(d, 32 + b); a21 = -readlane_f64(d, 33 + b); }
  DEVI void s2() {
    a22 = -readlane_f64(d, 34 + b);
    l20 = a20 * r0; l21 = __builtin_fma(-l20, l10, a21) * r1;
    r2 = rsqrt_nr(__builtin_fma(-l21, l21, __builtin_fma(-l20, l20, a22)));
  }
  DEVI void s3() {
    a30 = -readlane_f64(d, 48 + b); a31 = -readlane_f64(d, 49 + b); a32 = -readlane_f64(d, 50 + b);
    l30 = a30 * r0; l31 = __builtin_fma(-l30, l10, a31) * r1;
    l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, a32)) * r2;
  }
  DEVI void s4() {
    a33 = -readlane_f64(d, 51 + b);
    // the last pivot's 1/sqrt is not formed: it has one consumer, the lane's entry of row 3, which takes the estimate and the Newton
    // factor apart (s6) -- one product less behind the rsqrt, the last thing on the chain
    const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, a33)));
    rs3 = __builtin_amdgcn_rsq(d3);
    const double h3 = 0.5 * d3;
    e3n = __builtin_fma(-(h3 * rs3), rs3, 1.5);
#if CALICO_RSQRT_NEWTON_STEPS > 1
    { const double r = rs3 * e3n; rs3 = r; e3n = __builtin_fma(-(h3 * r), r, 1.5); }
#endif
  }
  DEVI void s5() {      // forward substitution L44 x = -e_lk (off the chain up to the last product)
    x0 = e0 * r0; x1 = __builtin_fma(-l10, x0, e1) * r1;
    x2 = __builtin_fma(-l21, x1, __builtin_fma(-l20, x0, e2)) * r2;
    x3a = __builtin_fma(-l32, x2, __builtin_fma(-l31, x1, __builtin_fma(-l30, x0, e3)));
    // the lane's row, picked by 0 / 1 factors (exact: one term per lane is not zero): rows 0..2 summed here, row 3's share as far as
    // it goes without the last pivot
    pre = __builtin_fma(m2, x2, __builtin_fma(m1, x1, m0 * x0));
    q3 = (m3 * x3a) * rs3;
  }
  DEVI double s6() {    // the lane's entry of -L44⁻¹ in the A-operand layout (row l16 < 4, k = lk; zero elsewhere)
    return __builtin_fma(q3, e3n, pre);
  }
  // The lane's row is picked by arithmetic on lane constants, not by a chain of selects on l16: the compiler turns that chain into a
  // switch -- three levels of exec-mask branches, twenty-odd instructions, in the middle of every step's chain (round 6, read off
  // the ISA) --, and the chain is issue-bound (~70 instructions at ~5 clocks): five FP64 instructions here against ten integer ones
  // for masks + and/or. (A failed pivot's NaN / Inf reaches every lane through the zero factors; such a factor is thrown away.)
  double m0, m1, m2, m3, pre, q3, rs3, e3n;
  DEVI void set_lane(int l16_) {
    l16 = l16_;
    m0 = l16_ == 0 ? 1.0 : 0.0; m1 = l16_ == 1 ? 1.0 : 0.0; m2 = l16_ == 2 ? 1.0 : 0.0; m3 = l16_ == 3 ? 1.0 : 0.0;
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
  }
};

#define CAL_SB() __builtin_amdgcn_sched_barrier(0)
#define CAL_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
// keeps all four result registers of a panel product allocated until here: only register 0 is used, and a dead register
// the compiler hands to the next VALU instruction costs that instruction the MFMA's whole latency in hazard wait states
#define CAL_KEEP(v) asm volatile("" : : "v"(v))

// The chief. In: rows 0..31 of A (LDS, row stride LD): the block, lower triangle (the upper one is not read).
// Out (WRITE_L = 1): rows 0..31: L, lower triangle (above the diagonal undefined); WRITE_L = 2: the lower triangle only --
// in place in a matrix whose strict upper triangle belongs to somebody else (the dense solve keeps L⁻ᵀ there).
// The spine of a 32x32 block as the chief holds it: tiles (0,0), (0,1), (1,1), negated, from the lower triangle of A (LDS, row stride LD).
DEVI void elim_load_spine(const double* A, int LD, int lane, f64x4& t00, f64x4& t01, f64x4& t11) {
  const int l16 = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = lk + 4 * r;
    const int hi = max(l16, c), lo = min(l16, c);
    t00[r] = -A[hi * LD + lo];
    t01[r] = -A[(16 + l16) * LD + c];
    t11[r] = -A[(16 + hi) * LD + 16 + lo];
  }
}
// The chief on a spine it already holds in registers (round 6, the tree levels' rolling chief: the wave that followed the block
// before with the rows of Bᵀ formed this block's diagonal -- D_next -= Z^BᵀZ^B, step by step, elim_follow_d -- in exactly this
// layout). WRITE_L as elim_chief; A is only touched when WRITE_L != 0.
template <int WRITE_L, bool TS = false, bool CC = false>
DEVI void elim_chief_reg(f64x4 t00, f64x4 t01, f64x4 t11, double* A, int LD, const ElimChannel ch, int lane, long long* ts = nullptr);
template <int WRITE_L, bool TS = false>
DEVI void elim_chief(double* A, int LD, const ElimChannel ch, int lane, long long* ts = nullptr) {
  f64x4 t00, t01, t11;
  elim_load_spine(A, LD, lane, t00, t01, t11);
  elim_chief_reg<WRITE_L, TS>(t00, t01, t11, A, LD, ch, lane, ts);
}
template <int WRITE_L, bool TS, bool CC>
DEVI void elim_chief_reg(f64x4 t00, f64x4 t01, f64x4 t11, double* A, int LD, const ElimChannel ch, int lane, long long* ts) {
  const int l16 = lane & 15, lk = lane >> 4;
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  PivotChain pc;
  pc.set_lane(l16);
  pc.e0 = lk == 0 ? -1.0 : 0.0; pc.e1 = lk == 1 ? -1.0 : 0.0; pc.e2 = lk == 2 ? -1.0 : 0.0; pc.e3 = lk == 3 ? -1.0 : 0.0;
  double* const pub = ch.buf + lane;
  f64x4 p0 = zero4, p1 = zero4;       // panel products (register 0: the step's columns of L for rows 0..15 / 16..31)
  // ---- columns 0..15. Per step: pivot chain -> w; p0 = P(t00), p1 = P(t01) back to back (both only need w); then the
  //      trailing updates t00 (what the next pivot block waits for), t01, t11. FP64 MFMAs and FP64 VALU work of one wave
  //      do not overlap on this part (the matrix instruction runs on the SIMD's FP64 lanes: 64 clocks each), so there is
  //      nothing to interleave -- the order just keeps dependent MFMAs apart. ----
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (TS) ts[u] = __builtin_readcyclecounter();
    pc.d = t00[u]; pc.b = 4 * u;
    pc.s0(); pc.s1(); pc.s2(); pc.s3(); pc.s4(); pc.s5();
    const double w = pc.s6();
    CAL_SB();
    elim_store(pub + elim_slot_w<CC>(u), w);
    p0 = CAL_MFMA(w, t00[u], zero4);
    p1 = CAL_MFMA(w, t01[u], zero4);
    if (u < 3) t00 = CAL_MFMA(p0[0], p0[0], t00);
    if (u < 3) t01 = CAL_MFMA(p0[0], p1[0], t01);
    t11 = CAL_MFMA(p1[0], p1[0], t11);
    if (!CC || u < 3) elim_store(pub + elim_slot_l0<CC>(u), p0[0]);      // (nobody reads l0 of step 3: the compact layout has no place for it)
    elim_store(pub + elim_slot_l1<CC>(u), p1[0]);
    if (WRITE_L) {
      if (WRITE_L == 1 || l16 >= 4 * u + lk) A[l16 * LD + 4 * u + lk] = p0[0];
      A[(16 + l16) * LD + 4 * u + lk] = p1[0];
    }
    CAL_SB();
    CAL_KEEP(p0); CAL_KEEP(p1);
  }
  // ---- columns 16..31: the spine is t11 alone. Followers need w and l1 (slot 2) of these steps. ----
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (TS) ts[4 + u] = __builtin_readcyclecounter();
    pc.d = t11[u]; pc.b = 4 * u;
    pc.s0(); pc.s1(); pc.s2(); pc.s3(); pc.s4(); pc.s5();
    const double w = pc.s6();
    CAL_SB();
    elim_store(pub + elim_slot_w<CC>(4 + u), w);
    if (u == 3 && !WRITE_L) break;                // (the last step: the followers only need w)
    p1 = CAL_MFMA(w, t11[u], zero4);
    if (u < 3) t11 = CAL_MFMA(p1[0], p1[0], t11);
    if (u < 3) elim_store(pub + elim_slot_l1<CC>(4 + u), p1[0]);
    if (WRITE_L == 1 || (WRITE_L == 2 && l16 >= 4 * u + lk)) A[(16 + l16) * LD + 16 + 4 * u + lk] = p1[0];
    CAL_SB();
    CAL_KEEP(p1);
  }
  CAL_KEEP(p0); CAL_KEEP(p1);
  if (TS) ts[8] = __builtin_readcyclecounter();
}

// A row tile a follower owns: sixteen rows against the block's 32 columns.
struct ElimTile {
  const double* in;      // entry (row i of the tile, column c of the block) at in[i * in_row + c * in_col]
  int in_row, in_col;
  double* out;           // where the result entry (i, c) goes: out[i * out_row + c * out_col]
  int out_row, out_col;
  int kind;              // 0: loaded from `in`; 1 / 2: identity rows against the block's columns 0..15 / 16..31 (-> L⁻ᵀ);
                         // 3: a single row (i = 0) loaded from `in`, the other fifteen are zero and nothing of them is stored
  double* diag;          // kinds 1 / 2, if not null: only the strict upper triangle of L⁻ᵀ goes to `out`, its diagonal to diag[i]
};

// A follower with NT row tiles (compile-time unrolled; the descriptors are wave-uniform).
// use_pre: the tiles come in registers -- pre0[q] / pre1[q] = the NEGATED entries of tile q against the block's columns 0..15 /
// 16..31 in the accumulator layout (register r of lane (l16, lk): row l16 of the tile, column lk + 4r) -- instead of being
// loaded from t[q].in: the tree levels' followers form the Schur update of their own input tiles in exactly that layout
// (bcr_level_kernel, look-ahead), so the updated tiles never go through LDS.
// (`use_pre` is a run-time, wave-uniform flag and not a template parameter: a second instantiation of the eight steps
//  doubled the loop-invariant output addresses the compiler keeps across the caller's loop -- into scratch.)
template <int NT, bool CC = false>
DEVI void elim_follow(const ElimTile (&t)[NT], const ElimChannel ch, int lane, bool use_pre = false, const f64x4* pre0 = nullptr, const f64x4* pre1 = nullptr) {
  const int l16 = lane & 15, lk = lane >> 4;
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  f64x4 x0[NT], x1[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    if (use_pre) { x0[q] = pre0[q]; x1[q] = pre1[q]; continue; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = lk + 4 * r;
      const double v0 = t[q].in[l16 * t[q].in_row + c * t[q].in_col], v1 = t[q].in[l16 * t[q].in_row + (16 + c) * t[q].in_col];
      const double id = l16 == c ? -1.0 : 0.0;
      const bool ld = t[q].kind == 0 || (t[q].kind == 3 && l16 == 0);
      x0[q][r] = ld ? -v0 : (t[q].kind == 1 ? id : 0.0);
      x1[q][r] = ld ? -v1 : (t[q].kind == 2 ? id : 0.0);
    }
  }
  const unsigned long long* const sub = reinterpret_cast<const unsigned long long*>(ch.buf) + lane;
  // what step s needs, the last-stored vector first (see ElimChannel); step s + 1 is requested before step s is worked
  // on, so that the LDS round trip runs beside the step's MFMAs (a request that came too early is repeated)
  unsigned long long v[3] = {0, 0, 0}, nv[3] = {0, 0, 0};
  auto request = [&](int s, unsigned long long (&d)[3]) {
    const bool need_l0 = s < 3, need_l1 = s < 7;
    if (need_l1) d[2] = __hip_atomic_load(sub + elim_slot_l1<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    d[0] = __hip_atomic_load(sub + elim_slot_w<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (need_l0) d[1] = __hip_atomic_load(sub + elim_slot_l0<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  request(0, v);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int J = s >> 2, u = s & 3;
    const bool need_l0 = s < 3, need_l1 = s < 7;
    while (__builtin_amdgcn_ballot_w64((need_l1 ? v[2] : v[0]) == kElimSentinel) != 0) { __builtin_amdgcn_s_sleep(1); request(s, v); }
    const double w = __longlong_as_double((long long)v[0]);
    const double l0 = need_l0 ? __longlong_as_double((long long)v[1]) : 0.0;
    const double l1 = need_l1 ? __longlong_as_double((long long)v[2]) : 0.0;
    if (s < 7) request(s + 1, nv);
    double lp[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const f64x4 p = CAL_MFMA(w, (J == 0 ? x0[q][u] : x1[q][u]), zero4);
      lp[q] = p[0];
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      {
        const int col = 4 * s + lk, row = l16 + (t[q].kind == 2 ? 16 : 0);      // (row: of L⁻ᵀ, for the identity tiles)
        double* dst = t[q].out + l16 * t[q].out_row + col * t[q].out_col;
        if (t[q].kind == 3) { if (l16 == 0) *dst = lp[q]; }
        else if ((t[q].kind == 1 || t[q].kind == 2) && t[q].diag) { if (col > row) *dst = lp[q]; else if (col == row) t[q].diag[l16] = lp[q]; }
        else *dst = lp[q];
      }
      if (need_l0) x0[q] = CAL_MFMA(l0, lp[q], x0[q]);
      if (need_l1 && !(J == 1 && u == 3)) x1[q] = CAL_MFMA(l1, lp[q], x1[q]);
    }
    v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2];
  }
}


// The rows of a follower's tile, negated, in the accumulator layout (what elim_follow loads itself): x0 / x1 = the tile against the
// block's columns 0..15 / 16..31. Apart from elim_follow_d so that the caller can tell the world "taken" between the loads and the steps.
DEVI void elim_load_rows(const double* in, int in_row, int in_col, int lane, f64x4& x0, f64x4& x1) {
  const int l16 = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = lk + 4 * r;
    x0[r] = -in[l16 * in_row + c * in_col];
    x1[r] = -in[l16 * in_row + (16 + c) * in_col];
  }
}
// The follower that becomes the next block's chief (round 6): its two row tiles are the rows of Bᵀ = T(next, this)ᵀ -- sixteen
// dimensions of the NEXT block each, against this block's 32 columns --, so its panel products are the step's four rows of Z^B, and
//     D_next -= Z^BᵀZ^B = Σ_steps lp lpᵀ
// accumulates, step by step, in the three spine tiles `n00`, `n01`, `n11` (negated, the chief's layout: elim_load_spine) straight
// from the panel products' result registers: the same 16x16x4 products in the same order as the tile products PᵀQ that used to
// form the update out of LDS behind a barrier (k = 4s .. 4s+3 per product, s ascending), so the sums are bit-identical. When the
// last step's products are through, the wave holds the next block's damped, updated diagonal and goes on as its chief
// (elim_chief_reg) -- nothing of it passes through LDS and nobody waits at a barrier.
// x0 / x1 [2]: the two row tiles (elim_load_rows); out[q]: where tile q's result entry (i, c) goes, out[q][i * out_row + c * out_col].
template <bool CC = false>
DEVI void elim_follow_d(f64x4 (&x0)[2], f64x4 (&x1)[2], double* out0, double* out1, int out_row, int out_col, const ElimChannel ch, int lane,
                        f64x4& n00, f64x4& n01, f64x4& n11) {
  const int l16 = lane & 15, lk = lane >> 4;
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  const unsigned long long* const sub = reinterpret_cast<const unsigned long long*>(ch.buf) + lane;
  unsigned long long v[3] = {0, 0, 0}, nv[3] = {0, 0, 0};
  auto request = [&](int s, unsigned long long (&d)[3]) {
    const bool need_l0 = s < 3, need_l1 = s < 7;
    if (need_l1) d[2] = __hip_atomic_load(sub + elim_slot_l1<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    d[0] = __hip_atomic_load(sub + elim_slot_w<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (need_l0) d[1] = __hip_atomic_load(sub + elim_slot_l0<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  request(0, v);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int J = s >> 2, u = s & 3;
    const bool need_l0 = s < 3, need_l1 = s < 7;
    while (__builtin_amdgcn_ballot_w64((need_l1 ? v[2] : v[0]) == kElimSentinel) != 0) { __builtin_amdgcn_s_sleep(1); request(s, v); }
    const double w = __longlong_as_double((long long)v[0]);
    const double l0 = need_l0 ? __longlong_as_double((long long)v[1]) : 0.0;
    const double l1 = need_l1 ? __longlong_as_double((long long)v[2]) : 0.0;
    if (s < 7) request(s + 1, nv);
    const f64x4 pa = CAL_MFMA(w, (J == 0 ? x0[0][u] : x1[0][u]), zero4);
    const f64x4 pb = CAL_MFMA(w, (J == 0 ? x0[1][u] : x1[1][u]), zero4);
    const double la = pa[0], lb = pb[0];
    // what the NEXT step's panel products wait for first, then the next block's diagonal (its tile (0,0) first: the pivot chain of
    // the block this wave is about to be the chief of starts there)
    if (need_l0) { x0[0] = CAL_MFMA(l0, la, x0[0]); x0[1] = CAL_MFMA(l0, lb, x0[1]); }
    if (need_l1 && !(J == 1 && u == 3)) { x1[0] = CAL_MFMA(l1, la, x1[0]); x1[1] = CAL_MFMA(l1, lb, x1[1]); }
    n00 = CAL_MFMA(la, la, n00);
    n01 = CAL_MFMA(la, lb, n01);
    n11 = CAL_MFMA(lb, lb, n11);
    {
      const int col = 4 * s + lk;
      out0[l16 * out_row + col * out_col] = la;
      out1[l16 * out_row + col * out_col] = lb;
    }
    CAL_KEEP(pa); CAL_KEEP(pb);
    v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2];
  }
}

// The owner of a block-row of a dense symmetric matrix following the elimination of an earlier diagonal block (round 6, the dense
// reduced solve's rolling form). Owner j holds, in registers, its rows of the block being eliminated (x0 / x1: two row tiles against
// the block's 32 columns) and its OWN diagonal block (n00, n01, n11: the chief's layout) and per step
//   * forms its four columns of Z (la, lb), stores them in place (out0 / out1: row stride ld, columns contiguous),
//   * updates the rest of its rows (x0 / x1) and its diagonal block  D_j -= Z Zᵀ  from the products' result registers,
//   * `cross` (owners two or more blocks behind the chief): also accumulates  Δ(j, l+1) = Z(j, l) Z(l+1, l)ᵀ  -- what this block does to
//     the rows the owner will follow the NEXT block with -- in nx0 / nx1; the other operand, the step's columns of Z of the owner
//     of block-row l+1, is read from where that owner stored it (c0p / c1p: its two row tiles) once its progress word says so;
//   * `dwave` (the owner of block-row l+1, the next chief): publishes its progress (prog = prog_base + step + 1) behind its stores.
// When the chief's last pivot is through, the owner of block-row l+1 holds D_{l+1} complete and goes on as the chief.
template <bool CC>
DEVI void elim_follow_owner(f64x4 (&x0)[2], f64x4 (&x1)[2], double* out0, double* out1, int ld, const ElimChannel ch, int lane,
                            f64x4& n00, f64x4& n01, f64x4& n11, bool cross, f64x4 (&nx0)[2], f64x4 (&nx1)[2],
                            const double* c0p, const double* c1p, bool dwave, int* prog, int prog_base) {
  const int l16 = lane & 15, lk = lane >> 4;
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  const unsigned long long* const sub = reinterpret_cast<const unsigned long long*>(ch.buf) + lane;
  unsigned long long v[3] = {0, 0, 0}, nv[3] = {0, 0, 0};
  auto request = [&](int s, unsigned long long (&d)[3]) {
    const bool need_l0 = s < 3, need_l1 = s < 7;
    if (need_l1) d[2] = __hip_atomic_load(sub + elim_slot_l1<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    d[0] = __hip_atomic_load(sub + elim_slot_w<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (need_l0) d[1] = __hip_atomic_load(sub + elim_slot_l0<CC>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  request(0, v);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int J = s >> 2, u = s & 3;
    const bool need_l0 = s < 3, need_l1 = s < 7;
    while (__builtin_amdgcn_ballot_w64((need_l1 ? v[2] : v[0]) == kElimSentinel) != 0) { __builtin_amdgcn_s_sleep(1); request(s, v); }
    const double w = __longlong_as_double((long long)v[0]);
    const double l0 = need_l0 ? __longlong_as_double((long long)v[1]) : 0.0;
    const double l1 = need_l1 ? __longlong_as_double((long long)v[2]) : 0.0;
    if (s < 7) request(s + 1, nv);
    // (the other owner's columns of this step are asked for now and looked at behind this step's own products)
    int pv = 0;
    double ca = 0.0, cb = 0.0;
    auto ask_cross = [&]() {
      pv = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      ca = c0p[l16 * ld + 4 * s + lk]; cb = c1p[l16 * ld + 4 * s + lk];
      asm volatile("" ::: "memory");
    };
    if (cross) ask_cross();
    const f64x4 pa = CAL_MFMA(w, (J == 0 ? x0[0][u] : x1[0][u]), zero4);
    const f64x4 pb = CAL_MFMA(w, (J == 0 ? x0[1][u] : x1[1][u]), zero4);
    const double la = pa[0], lb = pb[0];
    if (need_l0) { x0[0] = CAL_MFMA(l0, la, x0[0]); x0[1] = CAL_MFMA(l0, lb, x0[1]); }
    if (need_l1 && !(J == 1 && u == 3)) { x1[0] = CAL_MFMA(l1, la, x1[0]); x1[1] = CAL_MFMA(l1, lb, x1[1]); }
    n00 = CAL_MFMA(la, la, n00);
    n01 = CAL_MFMA(la, lb, n01);
    n11 = CAL_MFMA(lb, lb, n11);
    {
      // (behind the products: a store right behind the panel products waits out their latency with nothing else issued)
      const int col = 4 * s + lk;
      out0[l16 * ld + col] = la;
      out1[l16 * ld + col] = lb;
    }
    if (dwave) {
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_store(prog, prog_base + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
    }
    if (cross) {
      // the step's columns of Z of the next chief's rows: there once its progress word has passed this step (its stores come first)
      while (__builtin_amdgcn_readfirstlane(pv) < prog_base + s + 1) { __builtin_amdgcn_s_sleep(1); ask_cross(); }
      // tile (own row tile q, columns 0..15 / 16..31 of the next block) += Z_own Z_nextᵀ
      nx0[0] = CAL_MFMA(ca, la, nx0[0]); nx1[0] = CAL_MFMA(cb, la, nx1[0]);
      nx0[1] = CAL_MFMA(ca, lb, nx0[1]); nx1[1] = CAL_MFMA(cb, lb, nx1[1]);
    }
    CAL_KEEP(pa); CAL_KEEP(pb);
    v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2];
  }
}

}  // namespace cal
