// solve_dev.hpp — device helpers shared by the linear-solve kernel files (solve_kernels.hip, bcr_kernels.hip):
// reduce-buffer accessors, the LM bookkeeping that rides in other kernels, the in-wave 16-column panel factorisation
// and the 16x16 MFMA tile update.
#pragma once
#include <hip/hip_runtime.h>

#include "problem_dev.hpp"

#ifndef CALICO_RSQRT_NEWTON_STEPS
#define CALICO_RSQRT_NEWTON_STEPS 1
#endif

namespace cal {

#define DEVI __device__ __forceinline__

typedef double f64x4 __attribute__((ext_vector_type(4)));

// The reduce buffer that holds R(x): with speculative evaluation the Jacobian pass at the candidate point fills the
// other one and the control kernel swaps them when the step is accepted.
DEVI void use_current_R(SolveArgs& a) { if (a.r_stride && a.st->rcur) a.R += a.r_stride; }

DEVI double band_entry(const SolveArgs& a, int row, int col) {  // H(row, col), row >= col, inside the band
  const int ic = col / 6, cc = col % 6, ir = row / 6, rr = row % 6;
  const int d = ir - ic;
  if (d >= a.k) return 0.0;
  return a.R[a.off_B() + (size_t(ic) * a.k + d) * 36 + cc * 6 + rr];
}
DEVI double diag_entry(const SolveArgs& a, int j) {
  if (j < a.n_s()) return band_entry(a, j, j);
  const int i = j - a.n_s();
  return a.R[a.off_C() + size_t(i) * a.mc + i];
}


DEVI void publish(int* word, int value) { __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// Progress words of the streaming solve loop (host-mapped): [epoch << 20 | iterations the control stage is through with,
// epoch of the solve that has terminated].
DEVI void publish_progress(int* progress, const LmState* st, int seq) {
  if (st->terminated) publish(progress + 1, st->sink.epoch);
  publish(progress, (st->sink.epoch << 20) | seq);
}
// Results of a solve straight into host-visible pinned memory, by the stage that terminates it: a few kilobytes written
// over the fabric by one workgroup; the host returns as soon as it sees the terminated word, without waiting for the
// early-exit kernels of the iterations enqueued ahead. Call with all `nt` threads, between two barriers, after
// `terminated` was set; the caller publishes the progress words afterwards.
DEVI void publish_results_block(LmState* st, int tid, int nt) {
  const ResultSink k = st->sink;
  if (!k.state || st->published) return;
  const int n_state = int(sizeof(LmState) / sizeof(int)), n_row = int(sizeof(IterLog) / sizeof(int));
  const int rows = min(max(st->n_log, 0), k.rows);
  for (int i = tid; i < rows * n_row; i += nt) reinterpret_cast<int*>(k.log)[i] = reinterpret_cast<const int*>(k.src_log)[i];
  for (int i = tid; i < k.n_amb; i += nt) k.x[i] = k.src_x[i];
  for (int i = tid; i < n_state; i += nt) reinterpret_cast<int*>(k.state)[i] = reinterpret_cast<const int*>(st)[i];
  __threadfence_system();
}

DEVI void log_and_finalize(LmState* st, const LmOptionsDev& o, IterLog* log, int log_cap) {
  if (st->iteration > 0) { if (st->step_successful) st->num_successful++; else st->num_unsuccessful++; }
  const double row_cost = st->step_successful || st->iteration == 0 ? st->x_cost : (st->step_valid ? st->candidate_cost : st->x_cost);
  if (row_cost < st->min_cost) st->min_cost = row_cost;       // whether or not the row still fits the log
  st->last_logged_iteration = st->iteration;
  if (st->n_log < log_cap) {
    IterLog& r = log[st->n_log++];
    r.iteration = st->iteration; r.step_is_valid = st->step_valid; r.step_is_successful = st->step_successful; r.reserved = 0;
    r.cost = row_cost;
    r.cost_change = st->cost_change; r.gradient_max_norm = st->gradient_max_norm; r.step_norm = st->step_norm;
    r.relative_decrease = st->relative_decrease; r.trust_region_radius = st->radius;
  }
  if (st->iteration >= o.max_num_iterations) { st->terminated = 1; st->termination_type = 1; st->termination_reason = 1; return; }
  if (st->gradient_max_norm <= o.gradient_tolerance) { st->terminated = 1; st->termination_type = 0; st->termination_reason = 2; return; }
  if (st->radius < o.min_radius) { st->terminated = 1; st->termination_type = 0; st->termination_reason = 3; return; }
}

// After a Jacobian evaluation at x: x_cost, gradient norms |x - Plus(x,-g)|,
// Jacobi scaling at iteration 0, the iteration's log row.
// Call with ALL threads of the workgroup (256 or more): the first 256 do the work, every thread reaches every barrier.
DEVI void post_eval_body(SolveArgs a, const double* __restrict__ x, const BlockDev* __restrict__ blocks, int n_blocks,
                         const LmOptionsDev& o, IterLog* log, int log_cap, int first, int jacobi_scaling) {
  LmState* st = a.st;
  if (st->terminated || (!first && !st->need_jacobian)) return;
  use_current_R(a);
  __shared__ double s_max[256], s_sum[256];
  const int tid = threadIdx.x;
  const bool act = tid < 256;
  const int NT = a.NT();
  if (first && act) {
    // (behind the scale, 1 / s^2 = (1 + sqrt(diag))^2: the tree levels' chains damp their diagonal entries with products only)
    for (int j = tid; j < NT; j += 256) {
      const double q = 1.0 + sqrt(diag_entry(a, j));
      a.scale[j] = jacobi_scaling ? 1.0 / q : 1.0;
      a.scale[NT + j] = jacobi_scaling ? q * q : 1.0;
    }
  }
  double mx = 0.0, sm = 0.0;
  for (int b = act ? tid : n_blocks; b < n_blocks; b += 256) {
    const BlockDev B = blocks[b];
    const double* g = a.R + a.off_g() + B.tan_off;
    if (B.manifold == 1) {
      const double d0 = -g[0], d1 = -g[1], d2 = -g[2];
      const double nd = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
      if (nd > 0.0) {
        const double sd = sin(nd) / nd, cd = cos(nd);
        const double qx = sd * d0, qy = sd * d1, qz = sd * d2, qw = cd;
        const double* p = x + B.amb_off;  // x,y,z,w
        const double px = p[0], py = p[1], pz = p[2], pw = p[3];
        const double nw = qw * pw - qx * px - qy * py - qz * pz;
        const double nx = qw * px + qx * pw + qy * pz - qz * py;
        const double ny = qw * py + qy * pw + qz * px - qx * pz;
        const double nz = qw * pz + qz * pw + qx * py - qy * px;
        const double e[4] = {px - nx, py - ny, pz - nz, pw - nw};
        for (int i = 0; i < 4; ++i) { mx = fmax(mx, fabs(e[i])); sm += e[i] * e[i]; }
      }
    } else {
      for (int i = 0; i < B.size; ++i) { mx = fmax(mx, fabs(g[i])); sm += g[i] * g[i]; }
    }
  }
  if (act) { s_max[tid] = mx; s_sum[tid] = sm; }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_max[tid] = fmax(s_max[tid], s_max[tid + off]); s_sum[tid] += s_sum[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    st->x_cost = a.R[0];
    st->n_jac_evals += 1;
    st->gradient_max_norm = s_max[0];
    st->gradient_norm = sqrt(s_sum[0]);
    st->need_jacobian = 0;
    if (a.R[1] > 0.0) {  // a residual block failed to evaluate at an accepted point
      st->terminated = 1; st->termination_type = 2; st->termination_reason = first ? 10 : 11;
    } else {
      if (first) { st->initial_cost = st->x_cost; st->min_cost = st->x_cost; }
      log_and_finalize(st, o, log, log_cap);
    }
  }
  __syncthreads();
  if (st->terminated) {   // (uniform) the host stops enqueueing iterations and reads the results
    if (act) publish_results_block(st, tid, 256);
    __syncthreads();
    if (tid == 0) { st->published = 1; if (a.progress) publish(a.progress + 1, st->sink.epoch); }
  }
}


// This kernel's code, from here on, asked for as data (one word per 128-byte line), so that it stands in the XCD's L2 when the
// instruction fetch comes for it. Behind a kernel boundary a launch's code is not in the L2 any more (the evaluation and the gather
// move ~40 MB through the eight L2s per iteration), the sequencer's prefetch runs a line or two ahead, and straight-line code that
// runs once -- a single-block chief pass, a head -- then waits ~2k clocks per line nobody has fetched (round 6: per-step stamps of
// level 1's chief: one step of 2.4k clocks among steps of 0.55-0.8k, 2.2k in front of its first step). The caller keeps the returned
// value alive up to a point where it waits for its loads anyway (the loads of a wave return in order: asked for in front of loads
// that somebody waits for, they delay those). Reads past the kernel's end stay inside the code object's text segment.
DEVI int prefetch_code(int tid, int nthreads, int bytes) {
  const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc());
  int acc = 0;
  for (int q = tid * 128; q < bytes; q += nthreads * 128) acc += *reinterpret_cast<const int*>(pc + q);
  return acc;
}

DEVI double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d): hardware estimate (2^-24 relative, profiles/microbench/rsq_accuracy.hip) + Newton steps, no division and no
// sqrt call. Two steps give 1 ulp; one step gives 4e-15 relative, which is what the factorisation chains use: a
// Cholesky factor carries rounding errors of that order anyway (n·eps), and every pivot sits on a latency chain.
DEVI double rsqrt_nr(double d) {
  double r = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  r = r * (1.5 - h * r * r);
#if CALICO_RSQRT_NEWTON_STEPS > 1
  r = r * (1.5 - h * r * r);
#endif
  return r;
}

// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains vmcnt, which puts the full latency of
// every in-flight global prefetch / write-back on the per-step critical path of the sequential sweeps.
DEVI void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// Factor one 16-column panel (columns j0..j0+15 of rows j0..m) inside ONE wave; lane l owns rows j0 + l (+ 64·r).
// The latency chain runs through the diagonal only, and the diagonal entry of column jj+1 lives in lane jj+1, which
// holds everything it takes: pivot(jj+1) = a(jj+1, jj+1) - L(jj+1, jj)², both factors its own. So every lane follows
// its own would-be pivot `pown` (exact in the lane that matters), takes the reciprocal square root of it, and the
// chain per column is  rsqrt (v_rsq_f64 + one Newton step) -> one v_readlane pair (the pivot lane's 1/sqrt, now wave
// uniform) -> scale -> one FMA  ≈ 70 clocks (profiles/microbench/panel_latency.hip), with no second broadcast on it.
// Off the chain: columns jj+1 and jj+2 take column jj's contribution at once (multipliers by v_readlane, so that the
// diagonal entries two columns ahead are complete in time), the columns beyond one step later through a 64-double LDS
// buffer, as wave-uniform 16-byte reads requested before the chain starts. One wave issues a VALU instruction every
// 4+ clocks, so the count matters: v_readlane costs two instructions per multiplier, the LDS broadcast half an
// instruction. A bad pivot is not patched: it turns the factor into NaN (caught by the update stage); with PMIN the
// running minimum of the pivots is kept as well.
// DINV: also file the reciprocal diagonal (the dense reduced solve's backward sweep reads it; the tree solver does not).
template <int R, bool DINV = true, bool PMIN = true>
DEVI void panel_factor(double* A, int LD, double* dinv, double* bcast, int j0, int m, int w, int lane, double* pmin) {
  double av[R][16];
  int row[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    row[r] = j0 + lane + 64 * r;
    const double* src = A + min(row[r], m) * LD + j0;
#pragma unroll
    for (int c = 0; c < 16; ++c) av[r][c] = src[c];
  }
  double lprev[R], rs_keep = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) lprev[r] = 0.0;
  double pown = av[0][0];
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    // multipliers of column jj-1 for the columns from jj+2 on (written one step ago): requested before the pivot
    // chain starts, consumed after it
    double lb[16];
    if (jj > 0) {
      const double* br = bcast + ((jj - 1) & 1) * 64;
#pragma unroll
      for (int c = jj + 2; c < 16; ++c) lb[c] = br[c];
    }
    const double rs_own = rsqrt_nr(pown);
    const double rs = readlane_f64(rs_own, jj);
    const double piv_own = pown;
    double l[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { l[r] = av[r][jj] * rs; av[r][jj] = l[r]; }
    if (jj + 1 < 16) pown = __builtin_fma(-l[0], l[0], av[0][jj + 1]);    // the next pivot, in the lane that owns it
    double* bw = bcast + (jj & 1) * 64;
    bw[lane] = l[0];                                   // lanes 0..15 hold L(j0 + c, jj)
    if (PMIN) *pmin = fmin(*pmin, jj < w ? readlane_f64(piv_own, jj) : 1.0);     // off the chain
    if (jj + 1 < 16) {
      double lc = readlane_f64(l[0], jj + 1);
      if (R > 1) asm volatile("" : "+v"(lc));          // park it in a VGPR: as an SGPR pair it gets spilled between its uses
#pragma unroll
      for (int r = 0; r < R; ++r) av[r][jj + 1] = __builtin_fma(-l[r], lc, av[r][jj + 1]);
    }
    if (jj + 2 < 16) {
      double lc = readlane_f64(l[0], jj + 2);
      if (R > 1) asm volatile("" : "+v"(lc));
#pragma unroll
      for (int r = 0; r < R; ++r) av[r][jj + 2] = __builtin_fma(-l[r], lc, av[r][jj + 2]);
    }
    if (jj > 0) {
#pragma unroll
      for (int c = jj + 2; c < 16; ++c) {
#pragma unroll
        for (int r = 0; r < R; ++r) av[r][c] = __builtin_fma(-lprev[r], lb[c], av[r][c]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) lprev[r] = l[r];
    if (DINV) rs_keep = lane == jj ? rs_own : rs_keep;
  }
  if (DINV && lane < 16) dinv[j0 + lane] = rs_keep;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row[r] <= m) {
      double* dst = A + row[r] * LD + j0;
#pragma unroll
      for (int c = 0; c < 16; ++c) dst[c] = av[r][c];
    }
  }
}

// One 16×16 tile of the reduced matrix, block row I / block column c (16-blocks): D(I, c) -= Σ_q L(I, q) L(c, q)ᵀ over
// the factored panels q in [q0, q1), on the matrix cores. The tile is read and written once however many panels
// contribute (left-looking); rows past m are clamped for the operands and go to a dump word for the result.
DEVI void update_tile(double* A, int LD, int m, int I, int c, int q0, int q1, int lane, double* dump) {
  const int lr = lane & 15, lk = lane >> 4;
  const double* pa = A + min(16 * I + lr, m) * LD + lk;
  const double* pb = A + min(16 * c + lr, m) * LD + lk;
  double* pd[4];
  f64x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * I + lk + 4 * r;
    pd[r] = row <= m ? A + row * LD + 16 * c + lr : dump;
    acc[r] = *pd[r];
  }
  for (int q = q0; q < q1; ++q) {
    double va[4], vb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { va[kk] = -pa[16 * q + 4 * kk]; vb[kk] = pb[16 * q + 4 * kk]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kk], vb[kk], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) *pd[r] = acc[r];
}


}  // namespace cal
