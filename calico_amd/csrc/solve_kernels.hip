// solve_kernels.hip — normal-equation assembly and the LM step on gfx950.
//
// Every residual touches k consecutive control points plus a few calibration
// blocks, so JᵀJ is an arrowhead: a symmetric band B (6·n_cp rows, half
// bandwidth 6k-1) bordered by a dense strip E (6·n_cp × m) and a small dense
// corner C (m × m). The reference hands the same system to Ceres' DENSE_SCHUR
// (batch_optimizer.cpp:10-17,73), which factors it densely; here the band is
// eliminated first:
//   B = L Lᵀ (banded Cholesky),  Y = L⁻¹E,  S = C - YᵀY (dense, m×m),
//   S y_c = g_c - Yᵀ L⁻¹ g_s,    Lᵀ y_s = L⁻¹ g_s - Y y_c.
// The Levenberg–Marquardt damping and Ceres' Jacobi column scaling are folded
// into the diagonal, so the system solved is exactly Ceres':
//   (S_c H S_c + D²) y = S_c g,  D² = clamp(diag(S_c H S_c)) / radius,  delta = -S_c y.
#include <hip/hip_runtime.h>

#include "problem_dev.hpp"

namespace cal {

#define DEVI __device__ __forceinline__

DEVI double band_entry(const SolveArgs& a, int row, int col) {  // H(row, col), row >= col, inside the band
  const int ic = col / 6, cc = col % 6, ir = row / 6, rr = row % 6;
  const int d = ir - ic;
  if (d >= a.k) return 0.0;
  return a.R[a.off_B() + (size_t(ic) * a.k + d) * 36 + cc * 6 + rr];
}
DEVI double diag_entry(const SolveArgs& a, int j) {
  if (j < a.n_s()) return band_entry(a, j, j);
  const int i = j - a.n_s();
  return a.R[a.off_C() + size_t(i) * a.m + i];
}

// ---------------------------------------------------------------------------
// Gather: every entry of the reduce buffer is the sum of a fixed, host-built
// list of partial-block entries (CSR), summed in a fixed order => bitwise
// reproducible assembly with no atomics.
// ---------------------------------------------------------------------------
__global__ void gather_thin_kernel(double* __restrict__ R, const double* __restrict__ src, const int* __restrict__ out_idx,
                                   const int64_t* __restrict__ ptr, const int* __restrict__ idx, int n_out,
                                   const LmState* st, int need_flag) {
  if (st && (st->terminated || (need_flag && !st->need_jacobian))) return;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  double s = 0.0;
  for (int64_t q = ptr[o]; q < ptr[o + 1]; ++q) s += src[idx[q]];
  R[out_idx[o]] = s;
}
__global__ void gather_fat_kernel(double* __restrict__ R, const double* __restrict__ src, const int* __restrict__ out_idx,
                                  const int64_t* __restrict__ ptr, const int* __restrict__ idx, int n_out,
                                  const LmState* st, int need_flag) {
  if (st && (st->terminated || (need_flag && !st->need_jacobian))) return;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n_out) return;
  double s = 0.0;
  for (int64_t q = ptr[wave] + lane; q < ptr[wave + 1]; q += 64) s += src[idx[q]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) R[out_idx[wave]] = s;
}

// ---------------------------------------------------------------------------
// LM bookkeeping shared by the kernels below ([Ceres] trust_region_minimizer.cc
// FinalizeIterationAndCheckIfMinimizerCanContinue).
// ---------------------------------------------------------------------------
DEVI void log_and_finalize(LmState* st, const LmOptionsDev& o, IterLog* log, int log_cap) {
  if (st->iteration > 0) { if (st->step_successful) st->num_successful++; else st->num_unsuccessful++; }
  if (st->n_log < log_cap) {
    IterLog& r = log[st->n_log++];
    r.iteration = st->iteration; r.step_is_valid = st->step_valid; r.step_is_successful = st->step_successful; r.reserved = 0;
    r.cost = st->step_successful || st->iteration == 0 ? st->x_cost : (st->step_valid ? st->candidate_cost : st->x_cost);
    r.cost_change = st->cost_change; r.gradient_max_norm = st->gradient_max_norm; r.step_norm = st->step_norm;
    r.relative_decrease = st->relative_decrease; r.trust_region_radius = st->radius;
    if (r.cost < st->min_cost) st->min_cost = r.cost;
  }
  if (st->iteration >= o.max_num_iterations) { st->terminated = 1; st->termination_type = 1; st->termination_reason = 1; return; }
  if (st->gradient_max_norm <= o.gradient_tolerance) { st->terminated = 1; st->termination_type = 0; st->termination_reason = 2; return; }
  if (st->radius < o.min_radius) { st->terminated = 1; st->termination_type = 0; st->termination_reason = 3; return; }
}

// After a Jacobian evaluation at x: x_cost, gradient norms |x - Plus(x,-g)|,
// Jacobi scaling at iteration 0, the iteration's log row.
__global__ __launch_bounds__(256) void post_eval_kernel(SolveArgs a, const double* __restrict__ x,
                                                        const BlockDev* __restrict__ blocks, int n_blocks,
                                                        LmOptionsDev o, IterLog* log, int log_cap, int first,
                                                        int jacobi_scaling) {
  LmState* st = a.st;
  if (st->terminated || (!first && !st->need_jacobian)) return;
  __shared__ double s_max[256], s_sum[256];
  const int tid = threadIdx.x;
  const int NT = a.NT();
  if (first) {
    for (int j = tid; j < NT; j += 256) a.scale[j] = jacobi_scaling ? 1.0 / (1.0 + sqrt(diag_entry(a, j))) : 1.0;
  }
  double mx = 0.0, sm = 0.0;
  for (int b = tid; b < n_blocks; b += 256) {
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
  s_max[tid] = mx; s_sum[tid] = sm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_max[tid] = fmax(s_max[tid], s_max[tid + off]); s_sum[tid] += s_sum[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    st->x_cost = a.R[0];
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
}

// ---------------------------------------------------------------------------
// Build the damped working copies (one thread per entry).
// ---------------------------------------------------------------------------
__global__ void prepare_kernel(SolveArgs a, LmOptionsDev o) {
  const LmState* st = a.st;
  if (st->terminated) return;
  const int n_s = a.n_s(), W = a.W(), m = a.m, m1 = a.m + 1;
  const double radius = st->radius;
  const size_t nL = size_t(n_s) * W, nY = size_t(n_s) * m1, nS = size_t(m1) * m1;
  const size_t total = nL + nY + nS;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x) {
    if (i < nL) {
      const int c = int(i / W), t = int(i % W);
      const bool act = a.cp_active[c / 6] != 0;
      double v = 0.0;
      if (c + t < n_s) v = band_entry(a, c + t, c);
      if (t == 0) {
        if (!act) { v = 1.0; a.dadd[c] = 0.0; }
        else {
          const double s = a.scale[c];
          const double d = fmin(fmax(v * s * s, o.min_lm_diagonal), o.max_lm_diagonal) / (radius * s * s);
          a.dadd[c] = d; v += d;
        }
      } else if (!act || (c + t < n_s && !a.cp_active[(c + t) / 6])) v = 0.0;
      a.Lw[i] = v;
    } else if (i < nL + nY) {
      const size_t q = i - nL;
      const int c = int(q / m1), j = int(q % m1);
      const bool act = a.cp_active[c / 6] != 0;
      double v = 0.0;
      if (act) v = (j < m) ? a.R[a.off_E() + size_t(c) * m + j] : a.R[a.off_g() + c];
      a.Y[q] = v;
    } else {
      const size_t q = i - nL - nY;
      const int r = int(q / m1), cc = int(q % m1);
      double v = 0.0;
      if (r < m && cc < m) {
        v = a.R[a.off_C() + size_t(r) * m + cc];
        if (r == cc) {
          const double s = a.scale[n_s + r];
          const double d = fmin(fmax(v * s * s, o.min_lm_diagonal), o.max_lm_diagonal) / (radius * s * s);
          a.dadd[n_s + r] = d; v += d;
        }
      } else if (r == m && cc < m) v = a.R[a.off_g() + n_s + cc];   // right-hand side as an extra row
      else if (r < m && cc == m) v = a.R[a.off_g() + n_s + r];
      a.S[q] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Bordered banded Cholesky, right-looking, one workgroup. A ring of W+1 rows
// (band column + border row each) lives in LDS; finished rows stream out and
// fresh rows stream in while the window advances.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_cholesky_kernel(SolveArgs a) {
  LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double lds[];
  const int n = a.n_s(), W = a.W(), m1 = a.m + 1;
  const int RW = W + m1;          // ring row: [band column (W) | border row (m+1)]
  const int NR = W + 1;           // ring slots
  const int tid = threadIdx.x, nth = blockDim.x;
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  auto load_row = [&](int c) {
    double* row = lds + (c % NR) * RW;
    for (int i = tid; i < RW; i += nth) row[i] = (i < W) ? a.Lw[size_t(c) * W + i] : a.Y[size_t(c) * m1 + (i - W)];
  };
  for (int c = 0; c < W && c < n; ++c) load_row(c);
  __syncthreads();
  for (int c = 0; c < n; ++c) {
    double* rc = lds + (c % NR) * RW;
    const double p = rc[0];
    double l;
    if (!(p > 0.0) || !isfinite(p)) { if (tid == 0) s_fail = 1; l = 1.0; } else l = sqrt(p);
    const double inv = 1.0 / l;
    __syncthreads();  // everyone has read the pivot
    for (int i = tid; i < RW; i += nth) rc[i] = (i == 0) ? l : rc[i] * inv;
    if (c + W < n) load_row(c + W);   // free slot (c+W) % (W+1)
    __syncthreads();
    // trailing update + stream the finished row out
    const int tmax = min(W - 1, n - 1 - c);
    const int npairs = W * (W - 1) / 2;
    for (int q = tid; q < npairs; q += nth) {
      // q -> (t1, t2), 1 <= t1 <= t2 <= W-1
      int t1 = 1, rem = q;
      while (rem >= W - t1) { rem -= W - t1; ++t1; }
      const int t2 = t1 + rem;
      if (t2 <= tmax) lds[((c + t1) % NR) * RW + (t2 - t1)] -= rc[t2] * rc[t1];
    }
    for (int q = tid; q < tmax * m1; q += nth) {
      const int t = 1 + q / m1, j = q % m1;
      lds[((c + t) % NR) * RW + W + j] -= rc[t] * rc[W + j];
    }
    for (int i = tid; i < RW; i += nth) {
      if (i < W) a.Lw[size_t(c) * W + i] = rc[i]; else a.Y[size_t(c) * m1 + (i - W)] = rc[i];
    }
    __syncthreads();
  }
  if (tid == 0 && s_fail) st->chol_failed = 1;
}

// S -= YᵀY over the band rows (lower triangle incl. the right-hand-side row m).
__global__ void schur_kernel(SolveArgs a) {
  const LmState* st = a.st;
  if (st->terminated) return;
  const int m1 = a.m + 1, n = a.n_s();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m1 * m1) return;
  const int r = idx / m1, c = idx % m1;
  if (c > r || (r == a.m && c == a.m)) return;
  double s = 0.0;
  for (int q = 0; q < n; ++q) s += a.Y[size_t(q) * m1 + r] * a.Y[size_t(q) * m1 + c];
  a.S[size_t(r) * m1 + c] -= s;
}

// Dense Cholesky of the reduced system (rows 0..m-1) with the right-hand side
// carried as row m, then the backward substitution: y_c.
__global__ __launch_bounds__(256) void dense_cholesky_kernel(SolveArgs a, int use_lds) {
  LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double lds[];
  const int m = a.m, m1 = a.m + 1;
  const int tid = threadIdx.x, nth = blockDim.x;
  double* A = use_lds ? lds : a.S;
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  if (use_lds) { for (int i = tid; i < m1 * m1; i += nth) A[i] = a.S[i]; }
  __syncthreads();
  for (int j = 0; j < m; ++j) {
    const double p = A[j * m1 + j];
    double l;
    if (!(p > 0.0) || !isfinite(p)) { if (tid == 0) s_fail = 1; l = 1.0; } else l = sqrt(p);
    const double inv = 1.0 / l;
    __syncthreads();
    for (int i = j + tid; i <= m; i += nth) A[i * m1 + j] = (i == j) ? l : A[i * m1 + j] * inv;
    __syncthreads();
    const int rem = m - j;  // rows j+1..m
    for (int q = tid; q < rem * rem; q += nth) {
      const int i = j + 1 + q / rem, c = j + 1 + q % rem;
      if (c <= i && c < m) A[i * m1 + c] -= A[i * m1 + j] * A[c * m1 + j];
    }
    __syncthreads();
  }
  // row m now holds L⁻¹ b. Backward: Lᵀ y = that.
  for (int j = m - 1; j >= 0; --j) {
    if (tid == 0) A[m * m1 + j] = A[m * m1 + j] / A[j * m1 + j];
    __syncthreads();
    const double yj = A[m * m1 + j];
    for (int i = tid; i < j; i += nth) A[m * m1 + i] -= A[j * m1 + i] * yj;
    __syncthreads();
  }
  for (int i = tid; i < m; i += nth) a.y[a.n_s() + i] = A[m * m1 + i];
  if (tid == 0 && s_fail) st->chol_failed = 1;
}

// z = L⁻¹g_s - Y y_c ; then Lᵀ y_s = z by a backward band sweep (one wave).
__global__ __launch_bounds__(256) void back_substitute_kernel(SolveArgs a) {
  const LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double z[];  // n_s + W doubles
  const int n = a.n_s(), W = a.W(), m = a.m, m1 = a.m + 1;
  const int tid = threadIdx.x, nth = blockDim.x;
  const double* yc = a.y + n;
  for (int c = tid; c < n; c += nth) {
    const double* row = a.Y + size_t(c) * m1;
    double s = row[m];
    for (int j = 0; j < m; ++j) s -= row[j] * yc[j];
    z[c] = s;
  }
  for (int c = n + tid; c < n + W; c += nth) z[c] = 0.0;
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    // Lane t-1 (t = 1..W-1 <= 64) multiplies L(c+t, c) with y[c+t]; lane 0 finishes y[c].
    // The band columns are independent of the recurrence: prefetch them eight columns ahead.
    constexpr int PF = 8;
    double lcur[PF], dcur[PF], lnxt[PF], dnxt[PF];
    auto fetch = [&](int cbase, double* lv, double* dv) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int c = cbase - u;
        lv[u] = (c >= 0 && 1 + lane < W) ? a.Lw[size_t(c) * W + 1 + lane] : 0.0;
        dv[u] = (c >= 0) ? a.Lw[size_t(c) * W] : 1.0;
      }
    };
    fetch(n - 1, lnxt, dnxt);
    for (int cbase = n - 1; cbase >= 0; cbase -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) { lcur[u] = lnxt[u]; dcur[u] = dnxt[u]; }
      fetch(cbase - PF, lnxt, dnxt);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int c = cbase - u;
        if (c < 0) break;
        double prod = (1 + lane < W) ? lcur[u] * z[c + 1 + lane] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) prod += __shfl_xor(prod, off, 64);
        if (lane == 0) z[c] = (z[c] - prod) / dcur[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < n; c += nth) a.y[c] = z[c];
}

// delta = -y ; candidate = Plus(x, delta) ; model cost change ; step norms.
__global__ __launch_bounds__(256) void update_kernel(SolveArgs a, const double* __restrict__ x, double* __restrict__ x_cand,
                                                     const BlockDev* __restrict__ blocks, int n_blocks) {
  LmState* st = a.st;
  if (st->terminated) return;
  __shared__ double s_a[256], s_b[256], s_c[256];
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  const int NT = a.NT();
  double mcc = 0.0;
  for (int j = tid; j < NT; j += 256) {
    const double yj = a.y[j];
    if (!isfinite(yj)) s_bad = 1;
    mcc += 0.5 * yj * (a.R[a.off_g() + j] + yj * a.dadd[j]);
  }
  double sn = 0.0, cn = 0.0;
  for (int b = tid; b < n_blocks; b += 256) {
    const BlockDev B = blocks[b];
    const double* yb = a.y + B.tan_off;
    const double* p = x + B.amb_off;
    double* q = x_cand + B.amb_off;
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
      q[0] = nx; q[1] = ny; q[2] = nz; q[3] = nw;
      const double e[4] = {p[0] - nx, p[1] - ny, p[2] - nz, p[3] - nw};
      sn += e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
      cn += nx * nx + ny * ny + nz * nz + nw * nw;
    } else {
      for (int i = 0; i < B.size; ++i) {
        const double v = p[i] - yb[i];
        q[i] = v; const double e = p[i] - v; sn += e * e; cn += v * v;
      }
    }
  }
  s_a[tid] = mcc; s_b[tid] = sn; s_c[tid] = cn;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_a[tid] += s_a[tid + off]; s_b[tid] += s_b[tid + off]; s_c[tid] += s_c[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    st->model_cost_change = s_a[0];
    st->step_norm = sqrt(s_b[0]);
    st->candidate_cost = 0.0;
    st->cand_norm = sqrt(s_c[0]);
    if (s_bad || st->chol_failed) { st->step_valid = 0; }
    else st->step_valid = (s_a[0] > 0.0) ? 1 : 0;
  }
}

// Sum the per-item [cost, invalid] pairs of a cost-only evaluation into R2[0..1].
__global__ __launch_bounds__(256) void cost_reduce_kernel(const double* __restrict__ item_cost, int n_items, double* R2,
                                                          const LmState* st) {
  if (st && st->terminated) return;
  __shared__ double s_a[256], s_b[256];
  const int tid = threadIdx.x;
  double c = 0.0, v = 0.0;
  for (int i = tid; i < n_items; i += 256) { c += item_cost[2 * i]; v += item_cost[2 * i + 1]; }
  s_a[tid] = c; s_b[tid] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_a[tid] += s_a[tid + off]; s_b[tid] += s_b[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { R2[0] = s_a[0]; R2[1] = s_b[0]; }
}

// [Ceres] TrustRegionMinimizer: tolerance tests, step acceptance, radius update.
__global__ __launch_bounds__(256) void lm_control_kernel(LmState* st, LmOptionsDev o, const double* R2, double* x,
                                                         const double* x_cand, int n_amb, IterLog* log, int log_cap) {
  if (st->terminated) return;
  __shared__ int s_accept;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_accept = 0;
    const double cand_norm = st->cand_norm;
    st->iteration += 1;
    st->step_successful = 0; st->relative_decrease = 0.0; st->cost_change = 0.0;
    if (!st->step_valid) {
      // HandleInvalidStep
      st->step_norm = 0.0;
      if (++st->num_consecutive_invalid >= o.max_num_consecutive_invalid_steps) {
        st->terminated = 1; st->termination_type = 2; st->termination_reason = 12;
      } else {
        st->radius *= 0.5;
        st->chol_failed = 0;
        log_and_finalize(st, o, log, log_cap);
      }
    } else {
      st->num_consecutive_invalid = 0;
      const double candidate_cost = (R2[1] > 0.0) ? 1.7976931348623157e308 : R2[0];
      st->candidate_cost = candidate_cost;
      st->invalid_eval = R2[1] > 0.0;
      if (st->step_norm <= o.parameter_tolerance * (st->x_norm + o.parameter_tolerance)) {
        st->terminated = 1; st->termination_type = 0; st->termination_reason = 4;  // parameter tolerance
      } else {
        st->cost_change = st->x_cost - candidate_cost;
        if (fabs(st->cost_change) <= o.function_tolerance * st->x_cost) {
          st->terminated = 1; st->termination_type = 0; st->termination_reason = 5;  // function tolerance
        } else {
          st->relative_decrease = (candidate_cost >= 1.7976931348623157e308)
                                      ? -1.7976931348623157e308
                                      : (st->x_cost - candidate_cost) / st->model_cost_change;
          if (st->relative_decrease > o.min_relative_decrease) {
            s_accept = 1;
            st->step_successful = 1;
            st->need_jacobian = 1;
            st->x_norm = cand_norm;
            const double t = 2.0 * st->relative_decrease - 1.0;
            st->radius = st->radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            st->radius = fmin(o.max_radius, st->radius);
            st->decrease_factor = 2.0;
            // the log row of an accepted iteration is written by post_eval_kernel
          } else {
            st->radius = st->radius / st->decrease_factor;
            st->decrease_factor *= 2.0;
            log_and_finalize(st, o, log, log_cap);
          }
        }
      }
    }
  }
  __syncthreads();
  if (s_accept) {
    for (int i = tid; i < n_amb; i += 256) x[i] = x_cand[i];
  }
}

__global__ void init_state_kernel(LmState* st, double radius, double x_norm) {
  LmState s = {};
  s.radius = radius; s.decrease_factor = 2.0; s.x_norm = x_norm; s.need_jacobian = 1;
  s.min_cost = 1.7976931348623157e308;
  *st = s;
}

// ---- launch helpers ---------------------------------------------------------
void launch_gather(double* R, const double* src, const int* out_idx_thin, const int64_t* ptr_thin, const int* idx_thin,
                   int n_thin, const int* out_idx_fat, const int64_t* ptr_fat, const int* idx_fat, int n_fat,
                   const LmState* st, int need_flag, hipStream_t s) {
  if (n_thin > 0)
    hipLaunchKernelGGL(gather_thin_kernel, dim3((n_thin + 255) / 256), dim3(256), 0, s, R, src, out_idx_thin, ptr_thin,
                       idx_thin, n_thin, st, need_flag);
  if (n_fat > 0)
    hipLaunchKernelGGL(gather_fat_kernel, dim3((n_fat + 3) / 4), dim3(256), 0, s, R, src, out_idx_fat, ptr_fat, idx_fat,
                       n_fat, st, need_flag);
}
void launch_post_eval(const SolveArgs& a, const double* x, const BlockDev* blocks, int n_blocks, const LmOptionsDev& o,
                      IterLog* log, int log_cap, int first, int jacobi, hipStream_t s) {
  hipLaunchKernelGGL(post_eval_kernel, dim3(1), dim3(256), 0, s, a, x, blocks, n_blocks, o, log, log_cap, first, jacobi);
}
size_t band_cholesky_lds_bytes(const SolveArgs& a) { return size_t(a.W() + 1) * (a.W() + a.m + 1) * sizeof(double); }
size_t dense_cholesky_lds_bytes(const SolveArgs& a) { return size_t(a.m + 1) * (a.m + 1) * sizeof(double); }
hipError_t configure_solve_kernels(size_t band_lds, size_t dense_lds, size_t back_lds) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&band_cholesky_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, int(band_lds));
  if (e != hipSuccess) return e;
  if (dense_lds) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_cholesky_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, int(dense_lds));
    if (e != hipSuccess) return e;
  }
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&back_substitute_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(back_lds));
}
void launch_solve(const SolveArgs& a, const LmOptionsDev& o, const double* x, double* x_cand, const BlockDev* blocks,
                  int n_blocks, bool dense_in_lds, hipStream_t s) {
  const size_t total = size_t(a.n_s()) * a.W() + size_t(a.n_s()) * (a.m + 1) + size_t(a.m + 1) * (a.m + 1);
  const int pb = int((total + 255) / 256);
  hipLaunchKernelGGL(prepare_kernel, dim3(pb < 2048 ? pb : 2048), dim3(256), 0, s, a, o);
  hipLaunchKernelGGL(band_cholesky_kernel, dim3(1), dim3(256), band_cholesky_lds_bytes(a), s, a);
  const int m1 = a.m + 1;
  hipLaunchKernelGGL(schur_kernel, dim3((m1 * m1 + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(dense_cholesky_kernel, dim3(1), dim3(256), dense_in_lds ? dense_cholesky_lds_bytes(a) : 0, s, a,
                     dense_in_lds ? 1 : 0);
  hipLaunchKernelGGL(back_substitute_kernel, dim3(1), dim3(256), size_t(a.n_s() + a.W()) * sizeof(double), s, a);
  hipLaunchKernelGGL(update_kernel, dim3(1), dim3(256), 0, s, a, x, x_cand, blocks, n_blocks);
}
void launch_cost_reduce(const double* item_cost, int n_items, double* R2, const LmState* st, hipStream_t s) {
  hipLaunchKernelGGL(cost_reduce_kernel, dim3(1), dim3(256), 0, s, item_cost, n_items, R2, st);
}
void launch_control(LmState* st, const LmOptionsDev& o, const double* R2, double* x, const double* x_cand, int n_amb,
                    IterLog* log, int log_cap, hipStream_t s) {
  hipLaunchKernelGGL(lm_control_kernel, dim3(1), dim3(256), 0, s, st, o, R2, x, x_cand, n_amb, log, log_cap);
}
void launch_init_state(LmState* st, double radius, double x_norm, hipStream_t s) {
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, s, st, radius, x_norm);
}

}  // namespace cal
