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

#include <algorithm>
#include <cstdlib>
#include <string>

#include "problem_dev.hpp"
#include "solve_dev.hpp"

namespace cal {

// ---------------------------------------------------------------------------
// Gather: every entry of the reduce buffer is the sum of a fixed, host-built
// list of partial-block entries (CSR), summed in a fixed order => bitwise
// reproducible assembly with no atomics.
// ---------------------------------------------------------------------------
// One launch, two roles by workgroup: the first nb_fat workgroups take the fat outputs (> 48 sources, one wave
// each, the long ones first), the others the thin outputs (eight lanes each, strided over the sources); both end in
// a fixed-shape shuffle tree.
// What the control stage reads, requested before the workgroup's share of the gather so that the loads are back when
// the decision is due: the LM state (a working copy in LDS -- the scalar chain of the decision then runs on LDS instead
// of one global round trip per field), the tree solver's partial sums and the first candidate values of this thread.
struct ControlStage { double u0, u1, u2, u3; double xc[4]; const double* c01; };   // c01: the gather's outputs 0 and 1 (candidate cost, invalid count) in LDS
DEVI void control_prefetch(const LmState* st, const double* x_cand, int n_amb, LmState* s_st, ControlStage& pf);
DEVI void control_body(LmState* st, const LmOptionsDev& o, double* R2, double* x, const double* x_cand, int n_amb, IterLog* log,
                       int log_cap, const double* __restrict__ item_cost, int n_items, const double* Rbase, size_t r_stride,
                       int no_swap, int* progress, int seq, LmState* s_st, const ControlStage* staged = nullptr);
DEVI void publish(int* word, int value);
// ---------------------------------------------------------------------------
// Source lists built ON THE DEVICE (once per plan). For the band, the border and the spline part of the right-hand side
// -- 98 % of the outputs, 90 % of the list entries -- the sources of an output follow from its indices, because the band
// is uniform in time: output (control points a <= b, components r, c | calibration column tc) sums entry (i, j) of the
// expanded block of every cell (layout l, segment sg) whose segment covers a and b, i = 6(a - sg) + r, j = 6(b - sg) + c
// (or the local column of tc in layout l, or the block's last column for the right-hand side), in the order (layout,
// segment). The host therefore uploads three small tables instead of building the lists (4.4 of the 8.9 ms a fresh
// finalize took at configs[3]): the cells' blocks [n_lay][nseg], the local column of every calibration column in every
// layout [n_lay][m], the block side of every layout [n_lay] -- and three launches turn them into the same CSR lists the
// per-iteration gather reads (count, scan, fill). Outputs are numbered: the spline right-hand side, the band blocks (by distance
// from the diagonal, then in time), the border (row-major over the spline rows and the calibration columns). An output nothing
// contributes to (a control point beyond the trajectory's end in the band's layout) gets one source, a word that is
// always zero (`zero_slot`), so that every output has a list.
// ---------------------------------------------------------------------------
struct StructOut { int a, b, r, c, tc, rhs; size_t dst; };
DEVI StructOut struct_output(const GatherStruct& gs, int o) {
  const int NS = 6 * gs.n_cp, n_e = NS * gs.m, n_b = gs.n_cp * gs.k * 36;
  StructOut q;
  q.tc = -1; q.rhs = 0; q.c = 0;
  // (the spline right-hand side and the band first: they have the longest lists -- up to n_lay x k sources against k for most
  //  border entries --, and the workgroups that start last should be the light ones; it is also the order of R itself)
  if (o < NS) {
    const int ti = o;
    q.a = q.b = ti / 6; q.r = ti % 6; q.rhs = 1;
    q.dst = size_t(gs.off_g) + size_t(ti);
  } else if (o < NS + n_b) {
    // band blocks in two ranges -- distances d < d_split from the diagonal (long lists: eight lanes per output in the gather),
    // then the others (four lanes) --, and IN TIME inside a range: the gather deals contiguous pieces of a range to the XCDs,
    // and the outputs of one stretch of the trajectory read the same cells' blocks -- dealt by distance first, every XCD's L2
    // fetched every cell block (29 MB of fetches for 6 MB of blocks)
    const int e = o - NS, ds = gs.d_split, n_a = gs.n_cp * ds * 36;
    int bd, d;
    if (e < n_a) { bd = e / 36; q.a = bd / ds; d = bd - q.a * ds; }
    else { const int dr = gs.k - ds; bd = (e - n_a) / 36; q.a = bd / dr; d = ds + (bd - q.a * dr); }
    const int w = e - 36 * (e / 36);
    q.b = q.a + d;
    q.r = w / 6; q.c = w % 6;
    q.dst = size_t(gs.off_B) + size_t(q.a * gs.k + d) * 36 + size_t(w);
  } else {
    const int e = o - NS - n_b, ti = e / gs.m;
    q.tc = e - ti * gs.m;
    q.a = q.b = ti / 6; q.r = ti % 6;
    q.dst = size_t(gs.off_E) + size_t(e);
  }
  return q;
}
// calls f(src_index) for every source of output q, in list order; returns their number
template <class F>
DEVI int struct_sources(const GatherStruct& gs, const StructOut& q, F f) {
  const int n_lay = gs.n_lay, nseg = gs.nseg, m = gs.m, k = gs.k;
  const int* poff = gs.tab;
  const int* inv = gs.tab + n_lay * nseg;
  const int* n1s = inv + n_lay * m;
  if (q.b >= gs.n_cp) return 0;
  const int lo = max(0, q.b - (k - 1)), hi = min(q.a, nseg - 1);
  int n = 0;
  for (int l = 0; l < n_lay; ++l) {
    const int n1 = n1s[l];
    const int jc = q.tc >= 0 ? inv[l * m + q.tc] : 0;
    if (jc < 0) continue;
    for (int sg = lo; sg <= hi; ++sg) {
      const int po = poff[l * nseg + sg];
      if (po < 0) continue;
      int i = 6 * (q.a - sg) + q.r, j;
      if (q.tc >= 0) j = jc;
      else if (q.rhs) j = n1 - 1;
      else { j = 6 * (q.b - sg) + q.c; if (q.a == q.b && i > j) { const int t = i; i = j; j = t; } }
      f(po + tri_off(i, j, n1));      // (i <= j in every case: a spline row against a later spline column, a calibration column or the right-hand side)
      ++n;
    }
  }
  return n;
}
__global__ __launch_bounds__(256) void gather_lists_count_kernel(GatherStruct gs, int n_out, int* __restrict__ cnt, int* __restrict__ out_idx) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n_out) return;
  const StructOut q = struct_output(gs, o);
  const int n = struct_sources(gs, q, [](int) {});
  cnt[o] = n > 0 ? n : 1;
  out_idx[o] = int(q.dst);
}
// exclusive prefix sum of cnt[0..n) into ptr[0..n] (n is a few 10^4 .. 10^6): sums of 1024-element blocks, their scan by
// one workgroup, then every block scans its own elements on top of its offset -- coalesced reads and writes throughout
// (one workgroup walking the whole array with a stride per thread took 133 us for 82k outputs, 0.9 ms for 540k)
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(256) void gather_lists_block_sums_kernel(const int* __restrict__ cnt, int n, long long* __restrict__ bsum) {
  __shared__ long long sw[4];
  const int base = blockIdx.x * kScanBlock, tid = threadIdx.x;
  long long s = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int i = base + tid + 256 * u; s += i < n ? cnt[i] : 0; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((tid & 63) == 0) sw[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) bsum[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(1024) void gather_lists_scan_sums_kernel(long long* __restrict__ bsum, int nb, int64_t* __restrict__ ptr_total) {
  __shared__ long long part[1024];
  const int tid = threadIdx.x;
  long long carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {       // (one pass for up to 2^20 outputs)
    const int b = b0 + tid;
    const long long v = b < nb ? bsum[b] : 0;
    part[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const long long w = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += w;
      __syncthreads();
    }
    if (b < nb) bsum[b] = carry + part[tid] - v;      // exclusive
    const long long total = part[1023];
    __syncthreads();
    carry += total;
  }
  if (tid == 0) *ptr_total = carry;
}
__global__ __launch_bounds__(256) void gather_lists_block_scan_kernel(const int* __restrict__ cnt, int n, const long long* __restrict__ boff,
                                                                      int64_t* __restrict__ ptr) {
  __shared__ long long sw[4];
  const int base = blockIdx.x * kScanBlock, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // thread t owns elements base + 4t .. base + 4t + 3 (consecutive: the in-thread prefix is a register chain)
  int c[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int i = base + 4 * tid + u; c[u] = i < n ? cnt[i] : 0; }
  const long long mine = (long long)c[0] + c[1] + c[2] + c[3];
  long long incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const long long w = __shfl_up(incl, off, 64); if (lane >= off) incl += w; }
  if (lane == 63) sw[wave] = incl;
  __syncthreads();
  long long wave_off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_off += w < wave ? sw[w] : 0;
  long long run = boff[blockIdx.x] + wave_off + (incl - mine);
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int i = base + 4 * tid + u; if (i < n) ptr[i] = run; run += c[u]; }
}
__global__ __launch_bounds__(256) void gather_lists_fill_kernel(GatherStruct gs, int n_out, const int64_t* __restrict__ ptr, int* __restrict__ idx,
                                                                int zero_slot) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n_out) return;
  const StructOut q = struct_output(gs, o);
  int64_t w = ptr[o];
  const int n = struct_sources(gs, q, [&](int srci) { idx[w++] = srci; });
  if (n == 0) idx[w] = zero_slot;
}
// n_out outputs; idx must hold n_lay x k entries per output at most (the host sizes it by that bound); returns the total through
// ptr[n_out] (device)
void launch_gather_lists(const GatherStruct& gs, int n_out, int* cnt, int* out_idx, int64_t* ptr, int* idx, int zero_slot, long long* scratch,
                         hipStream_t s) {
  hipLaunchKernelGGL(gather_lists_count_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, gs, n_out, cnt, out_idx);
  const int nb = (n_out + kScanBlock - 1) / kScanBlock;
  hipLaunchKernelGGL(gather_lists_block_sums_kernel, dim3(nb), dim3(256), 0, s, cnt, n_out, scratch);
  hipLaunchKernelGGL(gather_lists_scan_sums_kernel, dim3(1), dim3(1024), 0, s, scratch, nb, ptr + n_out);
  hipLaunchKernelGGL(gather_lists_block_scan_kernel, dim3(nb), dim3(256), 0, s, cnt, n_out, scratch, ptr);
  hipLaunchKernelGGL(gather_lists_fill_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, gs, n_out, ptr, idx, zero_slot);
}

// `tail`: the LM control stage rides in the workgroup that finishes last (see ControlTail).
// U: sources per lane of a thin output (eight lanes): 6, or 12 for problems with outputs of 49..96 sources.
// Thin outputs [n_thin4, n_thin) have at most EIGHT sources each (the border: one layout's cells over the k segments of
// a control point) and take one lane instead of eight: an eighth of the threads for two thirds of the outputs; outputs
// [n_thin8, n_thin4) have at most 24 (the band's blocks away from the diagonal) and take four.
// FIXED (round 4): the thin outputs' source lists at a fixed stride per class (8·U / 24 / 8 entries, padded with the index of
// the word of the partials that is always zero) instead of CSR: a lane's indices follow from its output number alone, so the
// pointer load in front of them -- one of the three dependent round trips of this launch, ~1.9 us each behind a kernel
// boundary -- is gone, and so are the per-source bounds tests. Same sources in the same order: the sums are bit-identical.
template <int U, bool FIXED>
__global__ __launch_bounds__(256) void gather_kernel(double* __restrict__ R, const double* __restrict__ src,
                                                     const int* __restrict__ out_thin, const int64_t* __restrict__ ptr_thin,
                                                     const int* __restrict__ idx_thin, int n_thin, int n_thin8, int n_thin4,
                                                     const int* __restrict__ out_fat, const int64_t* __restrict__ ptr_fat,
                                                     const int* __restrict__ idx_fat, int n_fat, int nb_fat,
                                                     const double* __restrict__ cost_src, int n_cost,
                                                     const LmState* st, int need_flag, size_t other_stride, ControlTail tail, int xcd_map) {
  // (the list pointers of this thread's output do not depend on the state: requested before the flags are looked at)
  int64_t pre_q0 = 0, pre_q1 = 0;
  const int nb_thin8 = (n_thin8 + 31) / 32, nb_thin4 = (n_thin4 - n_thin8 + 63) / 64;
  int o_thin = 0;
  int cls = 8;       // lanes per output of this workgroup's thin outputs
  // xcd_map: workgroup p runs on XCD p % 8, each XCD has an L2 of its own, and the outputs are numbered in time inside a lane
  // class -- so every class is cut into eight contiguous pieces, one per XCD (the class's workgroups, padded to a multiple of
  // eight and starting at a multiple of eight, are dealt piece x = p % 8, position p / 8): the cells' blocks of one stretch of
  // the trajectory are then fetched by one L2 instead of by all eight.
  int tb_map = int(blockIdx.x) - 1 - nb_fat;
  bool idle = false;
  if (xcd_map && int(blockIdx.x) - 1 >= nb_fat) {
    const int pb = int(blockIdx.x) - ((1 + nb_fat + 7) & ~7);
    const int nb_thin1 = (n_thin - n_thin4 + 255) / 256;
    const int nbp8 = (nb_thin8 + 7) & ~7, nbp4 = (nb_thin4 + 7) & ~7;
    int pbc = pb, nbc = nb_thin8, base_l = 0;
    if (pb >= nbp8 + nbp4) { pbc = pb - nbp8 - nbp4; nbc = nb_thin1; base_l = nb_thin8 + nb_thin4; }
    else if (pb >= nbp8) { pbc = pb - nbp8; nbc = nb_thin4; base_l = nb_thin8; }
    const int j = (pbc & 7) * ((nbc + 7) >> 3) + (pbc >> 3);
    idle = pb < 0 || j >= nbc;
    tb_map = base_l + (idle ? 0 : j);
  }
  {
    const int bidp = int(blockIdx.x) - 1;
    if (bidp >= nb_fat && n_thin > 0) {
      const int tb = tb_map;
      int oc;
      if (tb < nb_thin8) { o_thin = (tb * int(blockDim.x) + int(threadIdx.x)) >> 3; oc = o_thin < n_thin8 ? o_thin : n_thin8 - 1; }
      else if (tb < nb_thin8 + nb_thin4) {
        cls = 4; o_thin = n_thin8 + (((tb - nb_thin8) * int(blockDim.x) + int(threadIdx.x)) >> 2); oc = o_thin < n_thin4 ? o_thin : n_thin4 - 1;
      } else {
        cls = 1; o_thin = n_thin4 + (tb - nb_thin8 - nb_thin4) * int(blockDim.x) + int(threadIdx.x); oc = o_thin < n_thin ? o_thin : n_thin - 1;
      }
      if (!FIXED) { pre_q0 = ptr_thin[oc]; pre_q1 = ptr_thin[oc + 1]; }
    }
  }
  // FIXED: idx_thin is the fixed-stride table [n_thin8][8U | n_thin4 - n_thin8][24 | n_thin - n_thin4][8]
  const size_t fbase4 = size_t(n_thin8) * (8 * U), fbase1 = fbase4 + size_t(n_thin4 - n_thin8) * 24;
  int fid[U > 8 ? U : 8];
  if (FIXED && int(blockIdx.x) - 1 >= nb_fat && n_thin > 0) {      // (requested before the state is looked at, like the pointers were)
    if (cls == 1) {
      const int oc = o_thin < n_thin ? o_thin : n_thin - 1;
#pragma unroll
      for (int u = 0; u < 8; ++u) fid[u] = idx_thin[fbase1 + size_t(oc - n_thin4) * 8 + u];
    } else if (cls == 4) {
      const int oc = o_thin < n_thin4 ? o_thin : n_thin4 - 1, sub = int(threadIdx.x) & 3;
#pragma unroll
      for (int u = 0; u < 6; ++u) fid[u] = idx_thin[fbase4 + size_t(oc - n_thin8) * 24 + sub + 4 * u];
    } else {
      const int oc = o_thin < n_thin8 ? o_thin : n_thin8 - 1, sub = int(threadIdx.x) & 7;
#pragma unroll
      for (int u = 0; u < U; ++u) fid[u] = idx_thin[size_t(oc) * (8 * U) + sub + 8 * u];
    }
  }
  if (st && (st->terminated || (need_flag && !st->need_jacobian))) {
    if (tail.enabled && tail.progress && st->terminated && blockIdx.x == 0 && threadIdx.x == 0) publish_progress(tail.progress, st, tail.seq);
    return;
  }
  if (idle) return;        // (padding of the XCD-aware dealing)
  __shared__ LmState s_st;
  __shared__ double s_c01[2];
  ControlStage pf;
  // Workgroup 0 sums the cost / invalid-count slots of the frames and work items (contiguous, interleaved [cost, invalid]:
  // no index list, one round trip with all 256 threads) into outputs 0 and 1 -- the longest sums of the launch -- and,
  // with `tail`, goes on to the LM control stage while the other workgroups assemble the normal equations.
  const bool owner = blockIdx.x == 0;
  if (owner && tail.enabled) { control_prefetch(st, tail.x_cand, tail.n_amb, &s_st, pf); pf.c01 = s_c01; }
  if (other_stride && st && st->rfill) R += other_stride;     // speculative: fill the buffer that does NOT hold R(x)
  if (owner) {
    const int tid = threadIdx.x;
    double c = 0.0, v = 0.0;
    for (int i = tid; i < n_cost; i += 256 * 4) {
      double cc[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = min(i + 256 * u, n_cost - 1); cc[u] = cost_src[2 * j]; vv[u] = cost_src[2 * j + 1]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i + 256 * u < n_cost) { c += cc[u]; v += vv[u]; }
    }
    c = wave_sum(c); v = wave_sum(v);
    __shared__ double s_w[2][4];
    if ((tid & 63) == 0) { s_w[0][tid >> 6] = c; s_w[1][tid >> 6] = v; }
    __syncthreads();
    if (tid == 0) {
      const double ct = ((s_w[0][0] + s_w[0][1]) + s_w[0][2]) + s_w[0][3], vt = ((s_w[1][0] + s_w[1][1]) + s_w[1][2]) + s_w[1][3];
      R[0] = ct; R[1] = vt; s_c01[0] = ct; s_c01[1] = vt;
    }
    if (tail.enabled) {
      __syncthreads();
      control_body(const_cast<LmState*>(st), tail.o, nullptr, tail.x, tail.x_cand, tail.n_amb, tail.log, tail.log_cap, nullptr, 0,
                   tail.Rbase, tail.r_stride, 0, tail.progress, tail.seq, &s_st, &pf);
    }
    return;
  }
  const int bid = int(blockIdx.x) - 1;
  // Both roles issue all index loads of a lane first and all value loads second: two memory round trips per output
  // instead of one dependent pair per source.
  if (bid < nb_fat) {
    const int wave_raw = (bid * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const bool live = wave_raw < n_fat;
    const int wave = live ? wave_raw : n_fat - 1;
    const int64_t q0 = ptr_fat[wave], q1 = ptr_fat[wave + 1];
    double s = 0.0;
    for (int64_t qb = q0; qb < q1; qb += 256) {     // four sources per lane and pass
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int64_t q = qb + lane + 64 * u; id[u] = idx_fat[q < q1 ? q : q1 - 1]; }
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[id[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += qb + lane + 64 * u < q1 ? v[u] : 0.0;
    }
    s = wave_sum(s);
    if (live && lane == 0) R[out_fat[wave]] = s;
  } else {
    if (FIXED) {
      if (cls == 1) {
        const int o = o_thin;
        const int dst = out_thin[o < n_thin ? o : n_thin - 1];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[fid[u]];
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
        if (o < n_thin) R[dst] = s;
        return;
      }
      if (cls == 4) {
        const int o = o_thin, sub = int(threadIdx.x) & 3;
        double v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = src[fid[u]];
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 6; ++u) s += v[u];
        s = row4_sum(s);
        if (o < n_thin4 && sub == 0) R[out_thin[o]] = s;
        return;
      }
      const int o = o_thin, sub = int(threadIdx.x) & 7;
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[fid[u]];
      // (the order of the CSR form: the first six per lane, then the others, added behind them)
      double s = 0.0, s2 = 0.0;
#pragma unroll
      for (int u = 0; u < 6; ++u) s += v[u];
#pragma unroll
      for (int u = 6; u < U; ++u) s2 += v[u];
      s += s2;
      s = row8_sum(s);
      if (o < n_thin8 && sub == 0) R[out_thin[o]] = s;
      return;
    }
    if (cls == 1) {      // one lane per output, at most eight sources, summed in list order
      const int o = o_thin;
      const int64_t q0 = pre_q0, q1 = pre_q1;
      int id[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t q = q0 + u; id[u] = idx_thin[q < q1 ? q : q1 - 1]; }
      const int dst = out_thin[o < n_thin ? o : n_thin - 1];
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[id[u]];
      double s = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += q0 + u < q1 ? v[u] : 0.0;
      if (o < n_thin) R[dst] = s;
      return;
    }
    if (cls == 4) {      // four lanes per output, at most 24 sources
      const int o = o_thin, sub = int(threadIdx.x) & 3;
      const int64_t q0 = pre_q0, q1 = pre_q1;
      int id[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int64_t q = q0 + sub + 4 * u; id[u] = idx_thin[q < q1 ? q : q1 - 1]; }
      double v[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) v[u] = src[id[u]];
      double s = 0.0;
#pragma unroll
      for (int u = 0; u < 6; ++u) s += q0 + sub + 4 * u < q1 ? v[u] : 0.0;
      s = row4_sum(s);
      if (o < n_thin4 && sub == 0) R[out_thin[o]] = s;
      return;
    }
    const int o = o_thin, sub = int(threadIdx.x) & 7;
    const bool live = o < n_thin8;
    const int64_t q0 = pre_q0, q1 = pre_q1;     // at most 8·U sources: U per lane
    // the first six per lane unconditionally; the others (U = 12) only where the list is that long -- most outputs
    // (the border's) have a handful of sources, and a wave whose groups are all short skips the second batch
    int id[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) { const int64_t q = q0 + sub + 8 * u; id[u] = idx_thin[q < q1 ? q : q1 - 1]; }
    double v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = src[id[u]];
    double s2 = 0.0;
    if constexpr (U > 6) {
      if (q1 - q0 > 48) {
        int id2[U - 6];
#pragma unroll
        for (int u = 6; u < U; ++u) { const int64_t q = q0 + sub + 8 * u; id2[u - 6] = idx_thin[q < q1 ? q : q1 - 1]; }
        double v2[U - 6];
#pragma unroll
        for (int u = 6; u < U; ++u) v2[u - 6] = src[id2[u - 6]];
#pragma unroll
        for (int u = 6; u < U; ++u) s2 += q0 + sub + 8 * u < q1 ? v2[u - 6] : 0.0;
      }
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 6; ++u) s += q0 + sub + 8 * u < q1 ? v[u] : 0.0;
    s += s2;
    s = row8_sum(s);
    if (live && sub == 0) R[out_thin[o]] = s;
  }
}

// ---------------------------------------------------------------------------
// LM bookkeeping shared by the kernels below ([Ceres] trust_region_minimizer.cc
// FinalizeIterationAndCheckIfMinimizerCanContinue).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void post_eval_kernel(SolveArgs a, const double* __restrict__ x,
                                                        const BlockDev* __restrict__ blocks, int n_blocks,
                                                        LmOptionsDev o, IterLog* log, int log_cap, int first,
                                                        int jacobi_scaling) {
  if (threadIdx.x == 0 && a.st->commit_pending) a.st->commit_pending = 0;   // see commit_kernel
  post_eval_body(a, x, blocks, n_blocks, o, log, log_cap, first, jacobi_scaling);
}

// ---------------------------------------------------------------------------
// Build the damped working copies (one thread per entry).
//   Lb[J][r][c] = H(6J + r, 6J + c), r = 0..6k-1, c = 0..5  (block column J of the band, lower part)
//   Y[c][j]     = E(c, j) for j < m, g_s(c) for j = m
//   S[r][c]     = C(r, c) (+ damping), row m / column m carry g_c
// ---------------------------------------------------------------------------
// With `with_post` the LAST workgroup does the bookkeeping of the step just accepted (post_eval_body: gradient norms,
// tolerance tests, log row) instead: it only reads R(x) and x, like the others, so it rides along instead of
// standing between the control kernel and the next linear solve.
__global__ __launch_bounds__(256) void prepare_kernel(SolveArgs a, LmOptionsDev o, int with_post, const double* __restrict__ x,
                                                      const BlockDev* __restrict__ blocks, int n_blocks, IterLog* log, int log_cap,
                                                      int jacobi_scaling) {
  const LmState* st = a.st;
  if (st->terminated) return;
  if (with_post && blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x == 0 && a.st->commit_pending) a.st->commit_pending = 0;   // see commit_kernel (several ranks)
    post_eval_body(a, x, blocks, n_blocks, o, log, log_cap, 0, jacobi_scaling);
    return;
  }
  const size_t n_prep_blocks = gridDim.x - (with_post ? 1 : 0);
  use_current_R(a);
  const int n_s = a.n_s(), W = a.W(), m = a.m, mc = a.mc, m1 = a.m + 1;
  const double radius = st->radius;
  const size_t nL = size_t(a.n_cp) * W * 6, nY = size_t(n_s) * m1, nS = size_t(m1) * m1;
  const size_t total = nL + nY + nS;
  // H between two tangent indices of the spline part (any order), zero outside the band
  auto hss = [&](int r, int c) { return r >= c ? band_entry(a, r, c) : band_entry(a, c, r); };
  // a control point takes part in the band unless it is unobserved or belongs to the separator
  auto band_act = [&](int J) { return a.cp_active[J] != 0 && !a.in_sep(6 * J); };
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += n_prep_blocks * blockDim.x) {
    if (i < nL) {
      const int J = int(i / (size_t(W) * 6)), rem = int(i % (size_t(W) * 6));
      const int r = rem / 6, cc = rem % 6;
      const int col = 6 * J + cc, row = 6 * J + r;
      const bool act = band_act(J);
      double v = 0.0;
      if (row < n_s) v = hss(row, col);
      if (row == col) {
        if (!act) { v = 1.0; if (!a.in_sep(col)) a.dadd[col] = 0.0; }
        else {
          const double s = a.scale[col];
          const double d = fmin(fmax(v * s * s, o.min_lm_diagonal), o.max_lm_diagonal) / (radius * s * s);
          a.dadd[col] = d; v += d;
        }
      } else if (!act || (row < n_s && !band_act(row / 6))) v = 0.0;
      a.Lb[i] = v;
    } else if (i < nL + nY) {
      const size_t q = i - nL;
      const int c = int(q / m1), j = int(q % m1);
      double v = 0.0;
      if (band_act(c / 6)) {
        if (j < mc) v = a.R[a.off_E() + size_t(c) * mc + j];
        else if (j < m) v = hss(c, 6 * a.sep_s + (j - mc));     // coupling of band row c to a separator column
        else v = a.R[a.off_g() + c];
      }
      a.Y[q] = v;
    } else {
      const size_t q = i - nL - nY;
      const int r = int(q / m1), cc = int(q % m1);
      double v = 0.0;
      if (r < m && cc < m) {
        if (r < mc && cc < mc) v = a.R[a.off_C() + size_t(r) * mc + cc];
        else if (r >= mc && cc >= mc) v = hss(6 * a.sep_s + (r - mc), 6 * a.sep_s + (cc - mc));
        else {   // separator row against calibration column
          const int t = 6 * a.sep_s + ((r >= mc ? r : cc) - mc), j = r >= mc ? cc : r;
          v = a.R[a.off_E() + size_t(t) * mc + j];
        }
        if (r == cc) {
          const int tj = a.border_tangent(r);
          const double s = a.scale[tj];
          const double d = fmin(fmax(v * s * s, o.min_lm_diagonal), o.max_lm_diagonal) / (radius * s * s);
          a.dadd[tj] = d; v += d;
        }
      } else if (r == m && cc < m) v = a.R[a.off_g() + a.border_tangent(cc)];   // right-hand side as an extra row
      else if (r < m && cc == m) v = a.R[a.off_g() + a.border_tangent(r)];
      a.S[q] = v;
    }
  }
}

// Bordered band Cholesky, blocked by control point (6 columns). Workgroup b factors the band (redundantly)
// together with border rows [b·16, (b+1)·16); the window of k block columns lives in an LDS ring.
// Step J:  panel X = A(:, J) L_JJ⁻ᵀ  |barrier|  trailing update A -= X Xᵀ  |barrier|, with one job per wave:
//  wave 0   panel by forward substitution, one lane per row (band rows, border rows, and six identity rows whose
//           solution is L_JJ⁻¹ for the back-substitution kernel); written out of place (Xbuf).
//  wave 1-2 streaming: coalesced write-back of block column J-1 and the ring refill (block J+k; J+k+2 requested);
//           global loads ride in registers for two steps, the barriers order LDS only, every load is unconditional.
//  wave 0-2 trailing update on the matrix cores (v_mfma_f64_16x16x4_f64, K = 6 padded to 8): 16-row tiles, one f64
//           per lane, operand and k-chunk -- the scalar 2×2-tile version was bound by LDS bandwidth.
//  wave 3   the latency chain, one step ahead and alone: X₁ (the six panel rows under the pivot) from L_JJ kept in
//           registers, the next pivot block minus X₁X₁ᵀ, its 6×6 Cholesky (six dependent rsqrt), publication of
//           L_{J+1,J+1} -- touching only private scratch, never waiting for the panel.
// One wave retires roughly one instruction every 5 clocks here, so the step is kept short by construction: every LDS
// offset and predicate is computed once per thread before the sweep (masked operands read a zero word, masked
// results go to a dump word, both inside each ring slot), divergent branches are avoided, and blocks past the end of
// the band are streamed in as zeros so that the short windows at the end need no special cases.
constexpr int kSlotPad = 16;     // per ring slot: zero words [0, 8), dump words [8, 16)

// Cholesky of a 6×6 block (lower, row-major 36) in every lane: L and the reciprocal diagonal. A non-positive or
// non-finite pivot is not patched: it turns the factor into NaN/Inf and is reported through *dmin / the caller.
// Right-looking: column j is scaled and the trailing triangle updated at once, so that the next pivot is ready one
// FMA after its multiplier -- the dependency chain per column is rsqrt + mul + fma instead of two dot products that
// grow with j (this routine sits on the per-step latency chain of the banded factorisation).
DEVI void chol6(const double* A, double L[6][6], double dinv[6], double* dmin) {
  double a[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k <= i; ++k) a[i][k] = A[i * 6 + k];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double d = a[j][j];
    *dmin = fmin(*dmin, d);
    const double inv = rsqrt_nr(d);
    L[j][j] = d * inv; dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) L[i][j] = a[i][j] * inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i)
#pragma unroll
      for (int k = j + 1; k <= i; ++k) a[i][k] -= L[i][j] * L[k][j];
  }
}

// (amdgpu_waves_per_eu(2): a 256-VGPR budget makes the compiler pick the VGPR form of the MFMAs; with the default
//  512-register budget it selects the AGPR form and copies every accumulator tile in and out.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void band_cholesky_kernel(SolveArgs a, int bs /* = kBorderSlice = 16 */) {
  LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double lds[];
  // blockIdx.y = band segment (two when a separator splits the band); everything below is relative to its first block
  const int jb = a.seg_begin(blockIdx.y);
  const int k = a.k, W = 6 * k, ncp = a.seg_end(blockIdx.y) - jb, m1 = a.m + 1;
  if (ncp <= 0) return;
  double* const Lb0 = a.Lb + size_t(jb) * W * 6;
  double* const Y0 = a.Y + size_t(6 * jb) * m1;
  double* const Linv0 = a.Linv + size_t(jb) * 36;
  const int j0 = blockIdx.x * bs;
  const int nb = max(1, min(bs, m1 - j0));
  const int nband = W * 6;
  const int SLP = nband + bs * 6;   // payload of a ring slot: band block [W][6] + border [bs][6]
  const int ZERO = SLP, DUMP = SLP + 8, LINV = SLP + kSlotPad;   // LINV: 36 doubles, used in Xbuf only
  const int SL = SLP + kSlotPad + 40;
  const int NSL = k + 2;            // active window (k) + two blocks in flight
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc16 = lane & 15, lk = lane >> 4;
  const int nelem = nband + nb * 6;
  constexpr int NE = 3;             // streamed elements per thread of waves 1..2: 384 >= 48*6 + 16*6
  double* Xbuf = lds + NSL * SL;    // [2][SL]: panel X (+ L⁻¹) of the current / previous step (slot layout), by step parity
  double* Lpiv = Xbuf + 2 * SL;     // [3][48]: L (36), reciprocal diagonal (6), by step mod 3
  double* W3 = Lpiv + 144;          // [80]: wave-3 scratch: X₁ (36), updated pivot block (36), dump (72..79)
  for (int i = tid; i < (NSL + 2) * SL + 224; i += 256) lds[i] = 0.0;
  // ---- streaming loads / factor write-back descriptors (a thread without an element duplicates another's) ----
  int l_off[NE], wb_l[NE];
  bool wb_piv[NE];
  const double* g_base[NE];
  double* wb_base[NE];
  size_t g_stride[NE], wb_stride[NE];
  const int stid = tid >= 64 && tid < 192 ? tid - 64 : 0;
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = stid + 128 * u;
    auto describe = [&](int ee, int* lo, size_t* goff, size_t* stride, bool* isy) {
      if (ee < nband) { *lo = ee; *goff = size_t(ee); *stride = size_t(nband); *isy = false; }
      else { const int q = ee - nband; const int c = q / nb, j = q % nb; *lo = nband + j * 6 + c; *goff = size_t(c) * m1 + j0 + j; *stride = size_t(6) * m1; *isy = true; }
    };
    size_t goff; bool isy;
    describe(e < nelem ? e : 0, &l_off[u], &goff, &g_stride[u], &isy);
    if (e >= nelem) l_off[u] = DUMP;
    g_base[u] = (isy ? Y0 : Lb0) + goff;
    const int ew = blockIdx.x == 0 ? e % nelem : nband + e % (nb * 6);   // write-back: workgroup 0 owns the band factor
    describe(ew, &wb_l[u], &goff, &wb_stride[u], &isy);
    wb_base[u] = (isy ? Y0 : Lb0) + goff;
    wb_piv[u] = ew < 36;                                                  // diagonal block: L comes from Lpiv, not from X
  }
  auto gload = [&](int J, double regs[NE]) {
    const int Jc = J < ncp ? J : ncp - 1;
#pragma unroll
    for (int u = 0; u < NE; ++u) regs[u] = g_base[u][size_t(Jc) * g_stride[u]];
  };
  // ---- panel rows of wave 0: band rows, border rows, identity rows (-> L⁻¹), idle ----
  const int nbr = W - 6;
  const int id_j = lane - (nbr + 16);                          // identity row index, valid in [0, 6)
  const int p_src = lane < nbr ? (6 + lane) * 6 : (lane < nbr + 16 ? nband + (lane - nbr) * 6 : ZERO);
  int p_dst[6];
#pragma unroll
  for (int c = 0; c < 6; ++c)
    p_dst[c] = lane < nbr + 16 ? p_src + c : (id_j < 6 ? LINV + c * 6 + id_j : DUMP + c);    // X(id row j, c) = L⁻¹(c, j)
  // ---- MFMA tiles of the trailing update: band tiles 0..nbt-1 (window rows 6..), tile nbt = border slice ----
  const int nbt = (nbr + 15) >> 4;
  const int n_bb = nbt * (nbt + 1) / 2, n_upd = n_bb + nbt;
  // slot-relative offset of X(row i of tile rt, q); ZERO when outside
  auto x_off = [&](int rt, int i, int q) -> int {
    if (q >= 6) return ZERO;
    if (rt == nbt) return i < nb ? nband + i * 6 + q : ZERO;
    const int wr = 16 * rt + i;
    return wr < nbr ? (6 + wr) * 6 + q : ZERO;
  };
  bool up_on[3];                   // waves 0..2: tiles t = wave, wave + 3, wave + 6
  int ua_off[3][2], ub_off[3][2], ut_off[3][4], ut_bc[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int t = wave + 3 * u;
    up_on[u] = wave < 3 && t < n_upd;
    int R = 0, C = 0;
    if (up_on[u]) {
      if (t < n_bb) { int rem = t; while (rem > R) { rem -= R + 1; ++R; } C = rem; }
      else { R = nbt; C = t - n_bb; }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { ua_off[u][kk] = x_off(R, lc16, 4 * kk + lk); ub_off[u][kk] = x_off(C, lc16, 4 * kk + lk); }
    // target of D(row = lk + 4r, col = lc16): window column wc lives in block column 1 + wc/6 at local column wc%6
    const int wc = 16 * C + lc16;
    const bool c_ok = wc < nbr;
    const int bc = c_ok ? wc / 6 : 0, ca = c_ok ? wc - 6 * bc : 0;
    ut_bc[u] = 1 + bc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lk + 4 * r, wr = 16 * R + i;
      bool ok; int lo;
      if (R == nbt) { ok = c_ok && i < nb; lo = nband + i * 6; }
      else { ok = c_ok && wr < nbr && wr >= wc && !(wr < 6 && wc < 6); lo = (wr - 6 * bc) * 6; }   // next pivot block: wave 3
      ut_off[u][r] = ok ? lo + ca : DUMP;
    }
  }
  // wave 3: lane (r, c) of the 6×6 blocks it works on; idle lanes compute on entry (0, 0) and write to the dump words
  const int pr = lane < 36 ? lane / 6 : 0, pcn = lane < 36 ? lane % 6 : 0;
  const int w3_p = lane < 36 ? 36 + lane : 72;
  const int w3_row = lane < 6 ? lane : 0;                      // X₁ row solved by this lane
  const int w3_xdst = lane < 6 ? lane * 6 : 72;
  __syncthreads();
  double regs[NE], regs2[NE];
  {
    // initial window: all k + 2 block loads are issued before the first LDS store (one memory round trip, not k)
    double first[8][NE];
#pragma unroll
    for (int J = 0; J < 8; ++J) if (J < k) gload(J, first[J]);
    gload(k, regs);       // blocks k and k+1 ride in registers
    gload(k + 1, regs2);
    if (wave == 1 || wave == 2) {
#pragma unroll
      for (int J = 0; J < 8; ++J) {
        if (J < k) {
          double* s0 = lds + (J % NSL) * SL;
#pragma unroll
          for (int u = 0; u < NE; ++u) s0[l_off[u]] = J < ncp ? first[J][u] : 0.0;
        }
      }
    }
  }
  long long tk0 = 0, tc[5] = {0, 0, 0, 0, 0};
  const bool dbg = CAL_DEV_TIMING(a.debug && blockIdx.x == 0 && (lane == 0));
#define TICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tc[i] += t_ - tk0; tk0 = t_; }
  // wave 3 keeps the current pivot factor in registers
  double Lc[6][6], dinv_c[6], dmin = 1.0;
  auto publish = [&](double* dst) {
    if (lane == 0) {
#pragma unroll
      for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int cc = 0; cc <= rr; ++cc) dst[rr * 6 + cc] = Lc[rr][cc];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) dst[36 + cc] = dinv_c[cc];
    }
  };
  // forward substitution x Lᵀ = a with the factor in registers / in LDS
  __syncthreads();
  if (wave == 3) { chol6(lds, Lc, dinv_c, &dmin); publish(Lpiv); }
  __syncthreads();
  int Jm = 0, J3 = 0;                           // J mod NSL, J mod 3
  for (int J = 0; J < ncp; ++J) {
    double* sj = lds + Jm * SL;
    double* xb = Xbuf + (J & 1) * SL;
    const double* piv = Lpiv + J3 * 48;
    double* piv_next = Lpiv + (J3 == 2 ? 0 : J3 + 1) * 48;
    const double* nxt = lds + (Jm + 1 < NSL ? Jm + 1 : 0) * SL;
    if (dbg) tk0 = __builtin_readcyclecounter();
    if (wave == 3) {
      // (3a) X₁ = A₁ L⁻ᵀ (rows 6..11 of block column J; lane r < 6 solves row r), then the pivot block of column
      //      J+1 minus X₁X₁ᵀ (lane (r, c))
      const double* a1 = sj + (6 + w3_row) * 6;
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) x[c] = a1[c];
#pragma unroll
      for (int c = 0; c < 6; ++c) {        // axpy form: one mul + one fma on the chain per column
        x[c] *= dinv_c[c];
#pragma unroll
        for (int q = c + 1; q < 6; ++q) x[q] -= x[c] * Lc[q][c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) W3[w3_xdst + c] = x[c];      // idle lanes: dump words 72..77
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: the writes above are visible to the reads below
      const double* xr = W3 + pr * 6;
      const double* xc = W3 + pcn * 6;
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) d += xr[q] * xc[q];
      W3[w3_p] = nxt[pr * 6 + pcn] - d;
      TICK(0)
      TICK(1)
    } else if (wave == 0) {
      // (1) panel by forward substitution: x_c = (a_c - Σ_{q<c} x_q L(c, q)) / L(c, c); lane = row
      double Lp[6][6], dv[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int cc = 0; cc < rr; ++cc) Lp[rr][cc] = piv[rr * 6 + cc];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) dv[cc] = piv[36 + cc];
      const double* ar = sj + p_src;
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) x[c] = ar[c] + (c == id_j ? 1.0 : 0.0);
#pragma unroll
      for (int c = 0; c < 6; ++c) {        // axpy form: one mul + one fma on the chain per column
        x[c] *= dv[c];
#pragma unroll
        for (int q = c + 1; q < 6; ++q) x[q] -= x[c] * Lp[q][c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) xb[p_dst[c]] = x[c];
      TICK(0)
      TICK(1)
    } else {
      // (2a) coalesced write-back of block column J-1 (panel X and L⁻¹ from Xbuf, L from Lpiv), one step late
      const int Jp = J > 0 ? J - 1 : 0;
      const double* sp = Xbuf + ((J & 1) ^ 1) * SL;
      const double* pp = Lpiv + (J3 == 0 ? 2 : J3 - 1) * 48;
      if (J > 0) {
#pragma unroll
        for (int u = 0; u < NE; ++u) wb_base[u][size_t(Jp) * wb_stride[u]] = wb_piv[u] ? pp[wb_l[u]] : sp[wb_l[u]];
        if (blockIdx.x == 0 && stid < 36) Linv0[size_t(Jp) * 36 + stid] = sp[LINV + stid];
      }
      // (2b) block J+k enters the ring (zeros past the end of the band), block J+k+2 is requested
      int sk = Jm + k; sk = sk >= NSL ? sk - NSL : sk;
      double* sn = lds + sk * SL;
      const bool on = J + k < ncp;
#pragma unroll
      for (int u = 0; u < NE; ++u) { sn[l_off[u]] = on ? regs[u] : 0.0; regs[u] = regs2[u]; }
      gload(J + k + 2, regs2);
      TICK(0)
      TICK(1)
    }
    lds_barrier();
    TICK(2)
    if (wave == 3) {
      // (3b) factor the next pivot block and publish it one step ahead
      if (J + 1 < ncp) { chol6(W3 + 36, Lc, dinv_c, &dmin); publish(piv_next); }
    } else {
      // (2) trailing update of the rest of the window on the matrix cores: D(R, C) -= X_R X_Cᵀ; all LDS reads first
      f64x4 acc[3];
      double* tb[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        int sidx = Jm + ut_bc[u]; sidx = sidx >= NSL ? sidx - NSL : sidx;
        tb[u] = lds + sidx * SL;
        if (!up_on[u]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[u][r] = tb[u][ut_off[u][r]];
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (!up_on[u]) continue;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xb[ua_off[u][kk]], xb[ub_off[u][kk]], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (!up_on[u]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[u][ut_off[u][r]] = acc[u][r];
      }
    }
    TICK(3)
    lds_barrier();
    TICK(4)
    Jm = Jm + 1 < NSL ? Jm + 1 : 0;
    J3 = J3 == 2 ? 0 : J3 + 1;
  }
  if (wave == 1 || wave == 2) {
    const int Jp = ncp - 1;
    const double* sp = Xbuf + (Jp & 1) * SL;
    const double* pp = Lpiv + (Jp % 3) * 48;
#pragma unroll
    for (int u = 0; u < NE; ++u) wb_base[u][size_t(Jp) * wb_stride[u]] = wb_piv[u] ? pp[wb_l[u]] : sp[wb_l[u]];
    if (blockIdx.x == 0 && stid < 36) Linv0[size_t(Jp) * 36 + stid] = sp[LINV + stid];
  }
  if (dbg) printf("band_cholesky cycles/step wave %d (0: panel | update; 1: stream | update; 3: X1+pivot update | factor+publish): A %lld  barrier %lld  B %lld  barrier %lld\n",
                  wave, (tc[0] + tc[1]) / ncp, tc[2] / ncp, tc[3] / ncp, tc[4] / ncp);
  if (wave == 3 && lane == 0) {
    double chk = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) chk += dinv_c[c];
    if (!(dmin > 0.0) || !isfinite(chk)) st->chol_failed = 1;
  }
}

// Sred = S - YᵀY, one 16×16 lower tile per workgroup, 64-row chunks staged in LDS with register prefetch.
constexpr int kSchurSlices = 2;   // K-slices when the in-LDS panel solver consumes the result (it sums them on load)
// `ks` workgroups share the rows (K dimension) of every tile: slice k writes its partial S/ks-th into
// Spart + k·(m+1)² (slice 0 carries S itself), and the consumer adds the slices up when it loads the matrix.
__global__ __launch_bounds__(256) void schur_kernel(SolveArgs a, int ks) {
  const LmState* st = a.st;
  if (st->terminated) return;
  // One 16×16 tile of YᵀY per workgroup and K-slice, on the matrix cores straight from global memory: lane l holds
  // Y[row k0 + (l >> 4)][16·t + (l & 15)], which is operand A of tile row t and operand B of tile column t of
  // v_mfma_f64_16x16x4_f64 -- coalesced 128-byte row segments, no LDS staging (the scalar version was bound by two
  // LDS reads per FMA). The four waves split the slice's rows and meet in LDS; fixed summation order.
  __shared__ double sacc[4][256];
  const int m1 = a.m + 1, n = a.n_s();
  const int tile = blockIdx.x / ks, slice = blockIdx.x % ks;
  int tr = 0, rem = tile;
  while (rem > tr) { rem -= tr + 1; ++tr; }
  const int tc = rem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc16 = lane & 15, lk = lane >> 4;
  const int rows_per = ((n + ks - 1) / ks + 15) & ~15;
  const int r_begin = slice * rows_per, r_end = min(n, r_begin + rows_per);
  const int per_wave = (((r_end - r_begin + 3) / 4 + 3) / 4) * 4;     // rows of one wave, a multiple of 4
  const int w_begin = r_begin + wave * per_wave, w_end = min(r_end, w_begin + per_wave);
  const int ca = 16 * tr + lc16, cb = 16 * tc + lc16;
  const double ma = ca < m1 ? 1.0 : 0.0, mb = cb < m1 ? 1.0 : 0.0;
  const double* pa = a.Y + min(ca, m1 - 1);
  const double* pb = a.Y + min(cb, m1 - 1);
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = w_begin; k0 < w_end; k0 += 32) {      // eight k-steps of loads in flight
    double va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = k0 + 4 * u + lk;
      const size_t ro = size_t(min(row, n - 1)) * m1;
      va[u] = pa[ro]; vb[u] = pb[ro];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double rm = k0 + 4 * u + lk < w_end ? 1.0 : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u] * (ma * rm), vb[u] * mb, acc, 0, 0, 0);
    }
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) + 4·reg
#pragma unroll
  for (int r = 0; r < 4; ++r) sacc[wave][(lk + 4 * r) * 16 + lc16] = acc[r];
  __syncthreads();
  {
    const int ti = tid >> 4, tj = tid & 15;
    const int r = tr * 16 + ti, c = tc * 16 + tj;
    const double sum = ((sacc[0][tid] + sacc[1][tid]) + sacc[2][tid]) + sacc[3][tid];
    if (r < m1 && c < m1 && c <= r)
      a.Spart[size_t(slice) * m1 * m1 + size_t(r) * m1 + c] = (slice == 0 ? a.S[size_t(r) * m1 + c] : 0.0) - sum;
  }
}

// Dense Cholesky of the reduced system (right-hand side carried as row m) and
// the backward substitution for y_c. One workgroup; one barrier per column.
__global__ __launch_bounds__(256) void reduced_solve_kernel(SolveArgs a, int use_lds) {
  LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double lds[];
  const int m = a.m, m1 = a.m + 1, n = a.n_s();
  const int LD = m1 | 1;  // odd row stride: conflict-free column walks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* A = use_lds ? lds : a.Swork;
  double* yv = A + size_t(m1) * LD;   // m1 doubles
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  for (int r = wave; r < m1; r += 4)
    for (int c = lane; c <= r; c += 64) A[r * LD + c] = a.Spart[size_t(r) * m1 + c];
  __syncthreads();
  {
    const int ti = tid >> 4, tj = tid & 15;
    double rs_prev = 0.0;
    for (int j = 0; j < m; ++j) {
      double p = A[j * LD + j];
      if (!(p > 0.0) || !isfinite(p)) { if (tid == 0) s_fail = 1; p = 1.0; }
      if (j > 0) { for (int i = j - 1 + tid; i <= m; i += 256) A[i * LD + (j - 1)] *= rs_prev; }  // lazy scaling of column j-1
      const double rs = rsqrt_nr(p);
      const double ip = rs * rs;
      for (int i = j + 1 + ti; i <= m; i += 16) {
        const double aij = A[i * LD + j] * ip;
        const int cmax = min(i, m - 1);
        for (int c = j + 1 + tj; c <= cmax; c += 16) A[i * LD + c] -= aij * A[c * LD + j];
      }
      rs_prev = rs;
      __syncthreads();
    }
    if (m > 0) { for (int i = m - 1 + tid; i <= m; i += 256) A[i * LD + (m - 1)] *= rs_prev; }
    __syncthreads();
  }
  // backward substitution Lᵀ y_c = (row m), one wave, dot-product form
  if (wave == 0) {
    for (int j = m - 1; j >= 0; --j) {
      double part = 0.0;
      for (int i = j + 1 + lane; i < m; i += 64) part += A[i * LD + j] * yv[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
      if (lane == 0) yv[j] = (A[m * LD + j] - part) / A[j * LD + j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  __syncthreads();
  for (int i = tid; i < m; i += 256) a.y[n + i] = yv[i];
  if (tid == 0 && s_fail) st->chol_failed = 1;
}

// Register-resident variant: thread (ti, tj) of a 16×16 grid owns the entries
// (ti + 16a, tj + 16b) of the reduced matrix for the whole factorisation; per
// column only the pivot column is broadcast through a double-buffered LDS
// vector, so a step costs one barrier, 2·NT LDS reads and ~NT²/2 register FMAs.
// The factor is kept row-major in Lrow (LDS or global); the backward substitution runs on one
// wave in axpy form: y_i = (b_i - acc_i)/L_ii, then acc_j += L_ij y_i for j < i, the pivot value
// travelling by v_readlane — no reduction on the dependency chain. Valid for m+1 <= 16·NT.
template <int NT, bool L_IN_LDS>
__global__ __launch_bounds__(256) void reduced_solve_reg_kernel(SolveArgs a) {
  LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double lds[];
  const int m = a.m, m1 = a.m + 1, n = a.n_s();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = tid >> 4, tj = tid & 15;
  constexpr int NP = 16 * NT;
  double* colbuf = lds;                 // [2][NP]
  double* dinv = lds + 2 * NP;          // [NP] reciprocal diagonal of the factor
  // row-major factor Lrow[i*NP + j], rows 0..m; the address space is a template parameter so that the
  // LDS variant compiles to ds_* instructions instead of flat ones
  double* Lrow;
  if constexpr (L_IN_LDS) Lrow = lds + 3 * NP; else Lrow = a.Swork;
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  double A[NT][NT];
#pragma unroll
  for (int aa = 0; aa < NT; ++aa)
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) {
      const int i = ti + 16 * aa, c = tj + 16 * bb;
      A[aa][bb] = (i <= m && c <= i && c < m1) ? a.Spart[size_t(i) * m1 + c] : 0.0;
    }
  const double dflag = (ti >= tj) ? 1.0 : 0.0;   // diagonal tiles: only c <= i
  const bool dbg = CAL_DEV_TIMING(a.debug && tid == 0);
  const long long t_begin = dbg ? __builtin_readcyclecounter() : 0;
  long long tph[4] = {0, 0, 0, 0}, tk = t_begin;
#define RTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  for (int j = 0; j < m; ++j) {
    double* cb = colbuf + (j & 1) * NP;
    const int bj = j >> 4;   // wave-uniform
    if (tj == (j & 15)) {
#pragma unroll
      for (int bb = 0; bb < NT; ++bb)
        if (bb == bj) {
#pragma unroll
          for (int aa = 0; aa < NT; ++aa) cb[ti + 16 * aa] = A[aa][bb];
        }
    }
    RTICK(0)
    __syncthreads();
    RTICK(1)
    double p = cb[j];
    if (!(p > 0.0) || !isfinite(p)) { if (tid == 0) s_fail = 1; p = 1.0; }
    const double rs = rsqrt_nr(p);
    double ri[NT], cj[NT];
#pragma unroll
    for (int aa = 0; aa < NT; ++aa) ri[aa] = cb[ti + 16 * aa] * rs;                                   // L(i, j)
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) cj[bb] = (tj + 16 * bb > j) ? cb[tj + 16 * bb] * rs : 0.0;        // L(c, j), c > j
#pragma unroll
    for (int aa = 0; aa < NT; ++aa) {
#pragma unroll
      for (int bb = 0; bb < aa; ++bb) A[aa][bb] -= ri[aa] * cj[bb];
      A[aa][aa] -= ri[aa] * cj[aa] * dflag;
    }
    RTICK(2)
    if (tj == (j & 15)) {   // stream the finished column out (row-major store)
#pragma unroll
      for (int aa = 0; aa < NT; ++aa) {
        const int i = ti + 16 * aa;
        if (i >= j && i <= m) Lrow[size_t(i) * NP + j] = ri[aa];
      }
      if (ti == 0) dinv[j] = rs;
    }
    RTICK(3)
  }
  __syncthreads();
  if constexpr (!L_IN_LDS) __threadfence();
  __syncthreads();
  const long long t_fact = dbg ? __builtin_readcyclecounter() : 0;
  if (wave == 0) {
    constexpr int NV = (NP + 63) / 64;
    double acc[NV], bq[NV], dv[NV], cur[NV], nxt[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int j = lane + 64 * u;
      acc[u] = 0.0;
      bq[u] = j < m ? Lrow[size_t(m) * NP + j] : 0.0;
      dv[u] = j < m ? dinv[j] : 0.0;
    }
    auto fetch = [&](int i, double v[NV]) {
#pragma unroll
      for (int u = 0; u < NV; ++u) { const int j = lane + 64 * u; v[u] = (i >= 0 && j < i) ? Lrow[size_t(i) * NP + j] : 0.0; }
    };
    fetch(m - 1, nxt);
    for (int i = m - 1; i >= 0; --i) {
#pragma unroll
      for (int u = 0; u < NV; ++u) cur[u] = nxt[u];
      fetch(i - 1, nxt);
      double cand = 0.0;
#pragma unroll
      for (int u = 0; u < NV; ++u) if ((i >> 6) == u) cand = (bq[u] - acc[u]) * dv[u];
      const double yi = readlane_f64(cand, i & 63);
#pragma unroll
      for (int u = 0; u < NV; ++u) acc[u] += cur[u] * yi;
      if (lane == 0) a.y[n + i] = yi;
    }
  }
  __syncthreads();
  if (dbg) printf("reduced_solve cycles/col: bcast-write %lld  barrier %lld  update %lld  store %lld | backward total %lld\n",
                  tph[0] / (m > 0 ? m : 1), tph[1] / (m > 0 ? m : 1), tph[2] / (m > 0 ? m : 1), tph[3] / (m > 0 ? m : 1),
                  (long long)__builtin_readcyclecounter() - t_fact);
  if (tid == 0 && s_fail) st->chol_failed = 1;
}

// Panel variant for m+1 <= 64·RPL: the augmented reduced matrix lives in LDS (row-major, odd stride) and is
// factored 16 columns at a time with look-ahead:
//   [block column p+1 receives the contribution of panel p, all waves] | barrier |
//   [wave 0: panel p+1]  ∥  [waves 1..3: block column p+2 receives the contributions of panels 0..p in one pass]
//   | barrier | ...
// so the latency chain of the panels (pivot -> rsqrt -> scale -> update, inside ONE wave, see panel_factor) never waits
// for the updates, which run on the matrix cores (v_mfma_f64_16x16x4_f64) and touch every tile of the matrix twice
// (left-looking) instead of once per panel. Row m is the right-hand side, so the forward substitution comes for free.
// Backward substitution by blocks of 16: an in-wave chain for the diagonal block, a parallel matrix-vector product
// for the rest (see below).
// `t0`, `nsl`: the kernel factors the trailing matrix from row/column t0 on (the blocked path hands over the last
// <= 128 rows once its panels have been eliminated; t0 = 0 otherwise) and adds up `nsl` K-slices on load.
template <int RPL>
__global__ __launch_bounds__(256) void reduced_solve_panel_kernel(SolveArgs a, int t0, int nsl) {
  LmState* st = a.st;
  if (st->terminated) return;
  extern __shared__ double lds[];
  const int M1 = a.m + 1;                        // row stride of the matrix in global memory
  const int m = a.m - t0, m1 = m + 1, n = a.n_s() + t0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = (m + 15) >> 4;
  const int LD = (16 * ((m1 + 15) / 16)) | 1;   // every 16-column panel stays inside its row
  double* A = lds;                              // [m1][LD] lower triangle
  double* dinv = lds + size_t(m1) * LD;         // [m + 16]
  double* bcast = dinv + m1 + 16;               // [2][64] column broadcast buffer of the panel wave / backward sweep
  double* dump = bcast + 128 + tid;             // [256] per-thread dump word
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  // load the lower triangle, summing the K-slices of the Schur complement: wave w takes rows w, w+4, ..., sixteen rows
  // per pass with every load of the pass issued before the first LDS store (a single workgroup pulls this matrix in,
  // so it is the number of loads in flight that matters: a dependent round trip costs about a microsecond)
  {
    const size_t mm = size_t(M1) * M1;
    const double* src0 = a.Spart + size_t(t0) * M1 + t0;
    for (int r0 = wave; r0 < m1; r0 += 64) {     // sixteen rows per wave and pass: two passes cover 128 rows
      double v[16][2];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = min(r0 + 4 * u, m1 - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = min(lane + 64 * h, r);
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < kSchurSlices; ++k)   // unconditional loads (a predicated load becomes a branch): slice index clamped, masked
            acc += src0[size_t(min(k, nsl - 1)) * mm + size_t(r) * M1 + c] * (k < nsl ? 1.0 : 0.0);
          v[u][h] = acc;
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + 4 * u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = lane + 64 * h;
          if (r < m1 && c <= r) A[r * LD + c] = v[u][h];
        }
      }
    }
  }
  __syncthreads();
  const bool dbg = CAL_DEV_TIMING(a.debug && (tid == 0 || tid == 64));
  long long tph[4] = {0, 0, 0, 0}, tk = dbg ? __builtin_readcyclecounter() : 0;
#define PTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  double pmin = 1.0;
  auto do_panel = [&](int j0) {
    const int w = min(16, m - j0);
    if (RPL > 1 && j0 + 64 <= m) panel_factor<RPL>(A, LD, dinv, bcast, j0, m, w, lane, &pmin);
    else panel_factor<1>(A, LD, dinv, bcast, j0, m, w, lane, &pmin);
  };
  // Left-looking with look-ahead, in 16-blocks: panel p = columns 16p.., row tiles I = 0..nrt-1 (the last one holds
  // the right-hand-side row m). Iteration p:
  //   phase 1, all waves : block column p+1 receives the contribution of panel p (it already holds those of 0..p-1)
  //   phase 2, wave 0    : panel p+1
  //            waves 1..3: block column p+2 receives the contributions of panels 0..p in one pass (each tile read and
  //                        written once)
  const int nrt = (m1 + 15) >> 4;
  if (wave == 0) do_panel(0);
  __syncthreads();
  for (int p = 0; p < nblk; ++p) {
    for (int I = p + 1 + wave; I < nrt; I += 4) update_tile(A, LD, m, I, p + 1, p, p + 1, lane, dump);
    PTICK(0)
    __syncthreads();
    PTICK(1)
    if (wave == 0) {
      if (p + 1 < nblk) do_panel(16 * (p + 1));
    } else {
      for (int I = p + 2 + (wave - 1); I < nrt; I += 3) update_tile(A, LD, m, I, p + 2, 0, p + 1, lane, dump);
    }
    PTICK(2)
    __syncthreads();
  }
  // backward substitution Lᵀ y = z (row m) by blocks of 16; thread t < 128 owns unknown t (waves 0 and 1 work, the
  // other two only keep the barriers company). Per block: the wave that owns its 16 unknowns solves the diagonal
  // block by an in-wave axpy chain on u = (z - pending sum) / L_jj, the value each lane would get if nothing more
  // were to come: step i broadcasts y_i = u(lane of i) by v_readlane and every lane still waiting takes
  // u -= (L(i, j) / L_jj) y_i -- one readlane pair and one FMA on the chain, the scaled multipliers being prepared
  // before it starts. One barrier; then every working thread adds L(blk, j)ᵀ y_blk to the pending sum of its own
  // unknown j < j0 -- a matrix-vector product, not a chain, with the entries of L fetched before the barrier.
  {
    const int j = tid;                         // unknown owned by this thread (m <= 127)
    const int jc = min(j, m - 1);
    double acc = 0.0, yk = 0.0;
    const double zq = j < m ? A[size_t(m) * LD + jc] : 0.0;
    const double dj = dinv[jc];
    for (int b = nblk - 1; b >= 0; --b) {
      const int j0 = 16 * b;
      double* ybuf = bcast + (b & 1) * 16;
      double lcol[16];                         // L(j0 + i, j), i = 0..15
#pragma unroll
      for (int i = 0; i < 16; ++i) lcol[i] = A[min(j0 + i, m - 1) * LD + jc];
      if (wave == (j0 >> 6)) {
        const int l0 = j0 & 63;                // lanes l0..l0+15 own the block
        const int i_own = lane - l0;           // row/column of this lane inside the block (valid when 0 <= i_own < 16)
        double ld[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ld[i] = (i_own >= 0 && i_own < i && j0 + i < m) ? -lcol[i] * dj : 0.0;
        double u = (zq - acc) * dj;            // 0 for the rows past m-1
#pragma unroll
        for (int i = 15; i >= 0; --i) {
          const double yi = readlane_f64(u, l0 + i);
          u += ld[i] * yi;
        }
        if (i_own >= 0 && i_own < 16) { yk = u; ybuf[i_own] = u; }
      }
      __syncthreads();
      if (wave < 2 && j < j0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += lcol[i] * ybuf[i];   // y of the rows past m-1 is 0
      }
    }
    if (j < m) a.y[n + j] = yk;
  }
  if (wave == 0 && lane == 0 && !(pmin > 0.0)) s_fail = 1;
  __syncthreads();
  PTICK(3)
  if (dbg) printf("reduced_solve_panel cycles (wave %d): column update %lld  barrier %lld  panel | trailing %lld  backward %lld\n", wave, tph[0], tph[1], tph[2], tph[3]);
#undef PTICK
  if (tid == 0 && s_fail) st->chol_failed = 1;
}

// ---------------------------------------------------------------------------
// Large reduced systems (m + 1 > 128: many cameras, free chart points): blocked right-looking Cholesky over all CUs,
// one launch per 32-column panel j. Workgroup (I, K), I >= K, owns the 64×64 tile of the trailing matrix at rows
// t0 + 64·I, columns t0 + 64·K (t0 = first row under the panel). Each of its four waves takes the pivot block A_jj
// in lanes 0-31 (one row per lane, 32 registers) and 32 rows of the panel under it in lanes 32-63, and runs the
// column Cholesky on the 64 rows at once, pivots and multipliers travelling by v_readlane: lanes 0-31 end up with
// L_jj (recomputed by every wave, which costs no time), lanes 32-63 with their rows of A_ij·L_jj⁻ᵀ. The four row
// blocks are those of tile rows I (waves 0, 1) and tile rows K (waves 2, 3); they meet in LDS and wave (a, b)
// subtracts P_{I,a}·P_{K,b}ᵀ from its 32×32 quarter of the tile, in place. The first tile column also files the
// panel into the factor L (Swork). The right-hand side rides as row m, so L(m, :) is the forward-substituted vector.
// `nsl`: K-slices of the Schur complement to add up on the first touch (panel 0 touches every entry).
constexpr int kRB = 32;
__global__ __launch_bounds__(256) void reduced_block_step_kernel(SolveArgs a, int j, int nsl) {
  LmState* st = a.st;
  if (st->terminated) return;
  __shared__ double sP[4][kRB][kRB + 1];
  __shared__ __attribute__((aligned(16))) double sCol[4][2][64];
  const int m = a.m, m1 = a.m + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = kRB * j, t0 = c0 + kRB;
  int I = 0, rem = blockIdx.x;
  while (rem > I) { rem -= I + 1; ++I; }
  const int K = rem;
  const int rI = t0 + 64 * I, rK = t0 + 64 * K;
  double* A = a.Spart;
  const size_t msq = size_t(m1) * m1;
  double* L = a.Swork;
  const bool dbg = CAL_DEV_TIMING(a.debug && blockIdx.x == 0 && j <= 1);
  long long tph[6] = {}, tk = dbg ? __builtin_readcyclecounter() : 0;
#define BTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  // ---- loads: this wave's 64 rows of panel j, and this lane's 4×4 piece of the tile ----
  const int rb = (wave < 2 ? rI : rK) + kRB * (wave & 1);
  const int myrow = lane < kRB ? c0 + lane : rb + (lane - kRB);
  const bool row_ok = myrow < m1;
  const double* src = A + size_t(min(myrow, m1 - 1)) * m1;
  double G[kRB];
#pragma unroll
  for (int c = 0; c < kRB; ++c) {
    const int col = c0 + c;                // may run past the row end: masked below, and Spart has slack behind it
    double v = src[col];
    for (int k = 1; k < nsl; ++k) v += src[size_t(k) * msq + col];
    const bool ok = row_ok && c0 + c < m1 && (lane >= kRB || c <= lane);
    G[c] = ok ? v : 0.0;
  }
  const int qa = wave >> 1, qb = wave & 1, r4 = lane >> 3, c4 = lane & 7;
  const int ur0 = rI + kRB * qa + 4 * r4, uc0 = rK + kRB * qb + 4 * c4;
  double pre[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const size_t o = size_t(min(ur0 + i, m1 - 1)) * m1 + uc0 + jj;
      double v = A[o];
      for (int k = 1; k < nsl; ++k) v += A[size_t(k) * msq + o];
      pre[i][jj] = v;
    }
  if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  BTICK(0)
  // ---- column Cholesky of the 64 rows ----
  // Column c: the unscaled column goes to a wave-private LDS vector; the pivot and the multiplier of column c+1 (the
  // latency chain) travel by v_readlane, the other multipliers come back as broadcast ds_read_b128 — one VALU
  // instruction per update instead of three.
  double (*cb)[64] = sCol[wave];
  double pmin = 1.0, psum = 0.0;
  // pivot path of column 0; inside the loop the one of column c+1 is started before the bulk of column c's updates,
  // so that the rsqrt chain runs under them
  double rs, t;
  {
    double p = readlane_f64(G[0], 0);
    p = c0 < m ? p : 1.0;                  // column m is the right-hand side, beyond it padding
    pmin = fmin(pmin, p); psum += p;       // a NaN pivot poisons psum, a non-positive one shows in pmin
    rs = rsqrt_nr(p);
    t = G[0] * (rs * rs);
  }
#pragma unroll
  for (int c = 0; c < kRB; ++c) {
    double* buf = cb[c & 1];
    buf[lane] = G[c];
    __builtin_amdgcn_wave_barrier();
    double rs_n = 1.0, t_n = 0.0;
    if (c + 1 < kRB) {
      G[c + 1] -= t * readlane_f64(G[c], c + 1);
      double p = readlane_f64(G[c + 1], c + 1);
      p = c0 + c + 1 < m ? p : 1.0;
      pmin = fmin(pmin, p); psum += p;
      rs_n = rsqrt_nr(p);
      t_n = G[c + 1] * (rs_n * rs_n);
    }
    // Multipliers of rows e, e+1 (constant bounds: everything unrolls). The four waves of the workgroup share one
    // LDS pipe, whose return path takes 4 clocks per broadcast double and wave; a v_readlane pair takes 8 clocks of
    // the wave's own SIMD. Alternating between the two balances the pipes.
    // The pins (empty asm over the updated columns and the pivot column) come after every group of 8 rows: without
    // them the compiler sinks every update to the column's first use (a left-looking schedule that keeps all 496
    // multipliers alive: 512 VGPRs and spills) or hoists all readlanes (SGPR spills through v_writelane).
    double gc = G[c];
    double2 U[kRB / 4];                     // all LDS reads of the column are issued before the first pin
#pragma unroll
    for (int e = 0; e < kRB; e += 4)
      if (e >= c + 2) U[e / 4] = *reinterpret_cast<const double2*>(buf + e);
#pragma unroll
    for (int e8 = 0; e8 < kRB; e8 += 8) {
      if (e8 + 7 < c + 2) continue;
#pragma unroll
      for (int e = e8; e < e8 + 8; e += 2) {
        if (e >= c + 2) {
          if ((e >> 1) & 1) {
            G[e] -= t * readlane_f64(gc, e);
            G[e + 1] -= t * readlane_f64(gc, e + 1);
          } else {
            G[e] -= t * U[e / 4].x;
            G[e + 1] -= t * U[e / 4].y;
          }
        } else if (e + 1 >= c + 2) {
          G[e + 1] -= t * readlane_f64(gc, e + 1);
        }
      }
      asm volatile("" : "+v"(gc), "+v"(G[e8]), "+v"(G[e8 + 1]), "+v"(G[e8 + 2]), "+v"(G[e8 + 3]), "+v"(G[e8 + 4]), "+v"(G[e8 + 5]), "+v"(G[e8 + 6]), "+v"(G[e8 + 7]));
    }
    G[c] = gc * rs;
    rs = rs_n; t = t_n;
  }
  BTICK(1)
  // ---- file the panel, exchange the row blocks ----
  if (lane >= kRB) {
#pragma unroll
    for (int c = 0; c < kRB; ++c) sP[wave][lane - kRB][c] = G[c];
    if (K == 0 && wave < 2 && row_ok) {   // masked entries go to a dump word past the factor: no branch per store
#pragma unroll
      for (int c = 0; c < kRB; ++c) L[c0 + c < m1 ? size_t(myrow) * m1 + c0 + c : msq + lane] = G[c];
    }
  } else if (blockIdx.x == 0 && wave == 0 && row_ok) {
#pragma unroll
    for (int c = 0; c < kRB; ++c) L[c <= lane ? size_t(myrow) * m1 + c0 + c : msq + lane] = G[c];
  }
  if (blockIdx.x == 0 && tid == 0 && (!(pmin > 0.0) || !isfinite(psum))) st->chol_failed = 1;
  __syncthreads();
  BTICK(2)
  // ---- tile update ----
  double acc[4][4] = {};
  const double (*PI)[kRB + 1] = sP[qa];
  const double (*PK)[kRB + 1] = sP[2 + qb];
#pragma unroll 8
  for (int k = 0; k < kRB; ++k) {
    double ra[4], rc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = PI[4 * r4 + i][k]; rc[i] = PK[4 * c4 + i][k]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] += ra[i] * rc[jj];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int ur = ur0 + i, uc = uc0 + jj;
      double* dst = (ur < m1 && uc <= ur) ? A + size_t(ur) * m1 + uc : L + msq + lane;
      *dst = pre[i][jj] - acc[i][jj];
    }
  BTICK(3)
  if (dbg && lane == 0) printf("reduced_block_step %d cycles (wave %d): loads %lld  factor %lld  file+barrier %lld  update+store %lld\n", j, wave, tph[0], tph[1], tph[2], tph[3]);
#undef BTICK
}

// Backward substitution Lᵀ y_c = L(m, :) for the blocked factor, one workgroup, panel by panel from the end:
// the half-wave whose threads own the panel's columns solves the 32×32 triangle in axpy form (one column of L_jj per
// lane, the solved unknown travelling by v_readlane), then every thread adds the panel's contribution to the
// pending sums of the columns it owns (column c belongs to thread c mod 256).
constexpr int kRBCols = 4;   // columns per thread: m <= 1024
// `np`: panels eliminated by the step kernels; the unknowns from kRB·np on were solved by the in-LDS kernel and enter
// the pending sums first.
__global__ __launch_bounds__(256) void reduced_block_back_kernel(SolveArgs a, int np) {
  LmState* st = a.st;
  if (st->terminated) return;
  __shared__ double sy[2][kRB];
  const int m = a.m, m1 = a.m + 1, n = a.n_s();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* L = a.Swork;
  double acc[kRBCols] = {};
  for (int i0 = kRB * np; i0 < m; i0 += kRB) {
    __syncthreads();
    if (tid < kRB) sy[0][tid] = i0 + tid < m ? a.y[n + i0 + tid] : 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRBCols; ++u) {
      const int col = tid + 256 * u;
      if (col < kRB * np) {
        double t[kRB];
#pragma unroll
        for (int i = 0; i < kRB; ++i) t[i] = L[size_t(min(i0 + i, m - 1)) * m1 + col];
        double sacc = 0.0;
#pragma unroll
        for (int i = 0; i < kRB; ++i) sacc += t[i] * sy[0][i];
        acc[u] += sacc;
      }
    }
  }
  __syncthreads();
  for (int jb = np - 1; jb >= 0; --jb) {
    const int c0 = kRB * jb;
    double* ybuf = sy[jb & 1];
    if (wave == ((c0 & 255) >> 6)) {
      const int l0 = c0 & 63, ci = lane - l0;       // this lane's column inside the panel (valid when 0 <= ci < 32)
      const int col = min(c0 + min(max(ci, 0), kRB - 1), m - 1);
      double Lc[kRB];
#pragma unroll
      for (int i = 0; i < kRB; ++i) Lc[i] = L[size_t(min(c0 + i, m - 1)) * m1 + col];
      const double zq = L[size_t(m) * m1 + col];
      double pend = 0.0;
#pragma unroll
      for (int u = 0; u < kRBCols; ++u) pend = (col >> 8) == u ? acc[u] : pend;
      double dj = 1.0;
#pragma unroll
      for (int i = 0; i < kRB; ++i) dj = ci == i ? Lc[i] : dj;
      dj = 1.0 / dj;
      double yk = 0.0;
#pragma unroll
      for (int i = kRB - 1; i >= 0; --i) {
        const double cand = (zq - pend) * dj;
        double yi = readlane_f64(cand, l0 + i);
        yi = c0 + i < m ? yi : 0.0;
        yk = ci == i ? yi : yk;
        pend += (ci >= 0 && ci < i) ? Lc[i] * yi : 0.0;
      }
      if (ci >= 0 && ci < kRB) {
        ybuf[ci] = yk;
        if (c0 + ci < m) a.y[n + c0 + ci] = yk;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRBCols; ++u) {
      const int col = tid + 256 * u;
      if (col < c0) {
        double t[kRB];
#pragma unroll
        for (int i = 0; i < kRB; ++i) t[i] = L[size_t(min(c0 + i, m - 1)) * m1 + col];
        double sacc = 0.0;
#pragma unroll
        for (int i = 0; i < kRB; ++i) sacc += t[i] * ybuf[i];
        acc[u] += sacc;
      }
    }
  }
}

// z = L⁻¹g_s - Y·y_c : one wave per band row, all CUs.  (y[0..n) used as z storage)
__global__ __launch_bounds__(256) void border_matvec_kernel(SolveArgs a) {
  const LmState* st = a.st;
  if (st->terminated) return;
  const int m = a.m, m1 = a.m + 1, n = a.n_s();
  const int lane = threadIdx.x & 63;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (c >= n) return;
  const double* row = a.Y + size_t(c) * m1;
  const double* yc = a.y + n;
  double part = 0.0;
  for (int j = lane; j < m; j += 64) part += row[j] * yc[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if (lane == 0) a.zbuf[c] = row[m] - part;
}

// delta = -y ; candidate = Plus(x, delta) ; model cost change ; step norms.
// (256 threads; called at the end of band_backsolve_kernel once the whole solution vector is in place)
// `part` of `n_parts`: with two band segments each back-substitution workgroup updates the control points of its own
// segment as soon as its sweep is done (part 0 also takes the separator and the calibration blocks, whose solution
// comes from the reduced solve) and leaves its partial sums in LmState; the control stage adds them up. No
// inter-workgroup synchronisation, hence no fences on the way.
DEVI void update_body(const SolveArgs& a_in, const double* __restrict__ x, double* __restrict__ x_cand,
                      const BlockDev* __restrict__ blocks, int n_blocks, int part, int n_parts) {
  SolveArgs a = a_in;
  use_current_R(a);
  LmState* st = a.st;
  __shared__ double s_a[256], s_b[256], s_c[256];
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  const int NT = a.NT();
  auto mine = [&](int tangent) {
    if (n_parts == 1) return true;
    const int owner = (tangent < a.n_s() && !a.in_sep(tangent) && tangent / 6 >= a.sep_s) ? 1 : 0;
    return owner == part;
  };
  double mcc = 0.0;
  for (int j = tid; j < NT; j += 256) {
    if (!mine(j)) continue;
    const double yj = a.y[a.y_index(j)];   // the separator part of the solution sits behind the calibration part
    if (!isfinite(yj)) s_bad = 1;
    mcc += 0.5 * yj * (a.R[a.off_g() + j] + yj * a.dadd[j]);
  }
  double sn = 0.0, cn = 0.0;
  for (int b = tid; b < n_blocks; b += 256) {
    const BlockDev B = blocks[b];
    if (!mine(B.tan_off)) continue;
    const double* yb = a.y + a.y_index(B.tan_off);
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
    st->upd_mcc[part] = s_a[0]; st->upd_sn[part] = s_b[0]; st->upd_cn[part] = s_c[0]; st->upd_bad[part] = s_bad;
    if (part == 0) { st->rfill = st->rcur ^ 1; st->upd_parts = n_parts; }
  }
}
// The update stage's partial sums -> model cost change, norms and the validity of the step (control stage, one thread).
DEVI void finish_update(LmState* st, int debug) {
  double mcc = 0.0, sn = 0.0, cn = 0.0;
  int bad = 0;
  for (int p = 0; p < st->upd_parts; ++p) { mcc += st->upd_mcc[p]; sn += st->upd_sn[p]; cn += st->upd_cn[p]; bad |= st->upd_bad[p]; }
  st->model_cost_change = mcc;
  st->step_norm = sqrt(sn);
  st->candidate_cost = 0.0;
  st->cand_norm = sqrt(cn);
  if (debug > 1) printf("update: model_cost_change %.6e  step_norm %.3e  non-finite %d  chol_failed %d\n", mcc, sqrt(sn), bad, st->chol_failed);
  st->step_valid = (bad || st->chol_failed) ? 0 : (mcc > 0.0 ? 1 : 0);
}

// Blocked backward band sweep Lᵀ y_s = z by ONE wave, in axpy form (no reductions on the chain):
//   step J:  s = z_J - P_J ;  y_J = L11⁻ᵀ s ;  for the k-1 earlier blocks B: P_B += L(J, B)ᵀ y_J.
// Lane (g, c) = (lane / 6, lane % 6), g < k-1, keeps in a register the pending sum P_B[c] of the one
// block B ≡ g (mod k-1) inside the window [J-(k-1), J-1]; when block J's turn comes its lanes form
// s, the six values travel by v_readlane (wave-uniform lane index), every lane gets y_J as scalars,
// and the update is six register FMAs. Band columns are prefetched four steps ahead.
template <int K> DEVI void band_backsolve_wave(const SolveArgs& a, int seg);
// One workgroup per band segment (blockIdx.x). With a single segment the other three waves of the workgroup wait at
// the barrier and then join the update of the candidate point (update_body) -- one launch less per iteration; with
// two segments the update needs both sweeps and runs as its own kernel.
template <int K>
__global__ __launch_bounds__(256) void band_backsolve_kernel(SolveArgs a, const double* __restrict__ x, double* __restrict__ x_cand,
                                                             const BlockDev* __restrict__ blocks, int n_blocks) {
  const LmState* st = a.st;
  if (st->terminated) return;
  if (threadIdx.x < 64) band_backsolve_wave<K>(a, blockIdx.x);
  __syncthreads();
  update_body(a, x, x_cand, blocks, n_blocks, blockIdx.x, a.n_seg());
}

template <int K>
DEVI void band_backsolve_wave(const SolveArgs& a, int seg) {
  constexpr int W = 6 * K, G = K - 1, nband = W * 6, PF = 4, NSLOT = PF + 1;
  const int jb = a.seg_begin(seg);
  const int ncp = a.seg_end(seg) - jb;       // blocks of this segment; all indices below are relative to jb
  if (ncp <= 0) return;
  const int lane = threadIdx.x;
  const int g = lane / 6, c = lane % 6;
  const bool worker = g < G;
  const double* __restrict__ z = a.zbuf + 6 * jb;   // from border_matvec_kernel; distinct from the output so loads can stay in flight
  double* __restrict__ yout = a.y + 6 * jb;
  const double* __restrict__ Lbp = a.Lb + size_t(jb) * nband;
  const double* __restrict__ Lip = a.Linv + size_t(jb) * 36;
  // block handled by this lane at step J: B = J - d, d in [1, G], B ≡ g (mod G)
  auto dist = [&](int J) { int d = (J - g) % G; if (d < 0) d += G; return d == 0 ? G : d; };
  // Every load of the sweep is unconditional (addresses clamped into the arrays) and issued by fetch(), PF steps
  // ahead: conditional loads compile to exec-masked branches and the waitcnt pass then falls back to vmcnt(0),
  // which collapses the prefetch distance to one step. Values fetched for B < 0 / idle lanes are never consumed.
  // The ring has PF+1 slots and the loop is unrolled by PF+1: step u consumes slot u and refills the slot consumed
  // one step earlier (dead by then), so no register copy -- and no vmcnt(0) -- is needed at the loop back-edge.
  double lring[NSLOT][6];
  double iv[NSLOT][6];  // lane c' < 6: Linv[J][q][c'], q = 0..5 (column c' of the inverse = row of its transpose)
  double zring[NSLOT];  // z of block J - G, picked up by the lanes whose block is consumed at step J
  const int lane6 = lane < 6 ? lane : 0;
  auto fetch = [&](int J, double lv[6], double ivv[6], double& zn) {
    const int Jc = J > 0 ? J : 0;
    const int d = dist(Jc);
    const int B = Jc - d, Bc = B > 0 ? B : 0;
    const double* lp = Lbp + size_t(Bc) * nband + (6 * d) * 6 + c;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) lv[rr] = lp[rr * 6];
    const double* ip = Lip + size_t(Jc) * 36 + lane6;
#pragma unroll
    for (int q = 0; q < 6; ++q) ivv[q] = ip[q * 6];
    const int jn = Jc - G;
    zn = z[6 * (jn > 0 ? jn : 0) + c];
  };
  double P = 0.0;
  // z of the first block this lane will consume: the largest J' <= ncp-1 with J' ≡ g (mod G)
  int jz = ncp - 1 - (((ncp - 1 - g) % G + G) % G);
  double zc = z[6 * (jz > 0 ? jz : 0) + c];
  // pin this load before the prefetches: otherwise the loop-header waitcnt merges to vmcnt(0) on every trip
  asm volatile("" : "+v"(zc) : : "memory");
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(ncp - 1 - u, lring[u], iv[u], zring[u]);
  for (int J0 = ncp - 1; J0 >= 0; J0 -= NSLOT) {
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
      const int J = J0 - u;   // J < 0 in the last group: a harmless dummy step on clamped loads, no store
      double lcur[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) lcur[rr] = lring[u][rr];
      double icur[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) icur[q] = iv[u][q];
      const double znext = zring[u];
      const int rf = (u + PF) % NSLOT;
      fetch(J - PF, lring[rf], iv[rf], zring[rf]);
      const int gj = (J + G * NSLOT) % G;
      // s_c on lanes (gj, c)
      const double sc = zc - P;
      double s[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) s[q] = readlane_f64(sc, 6 * gj + q);
      // consumed: start the pending sum of block J - G
      P = g == gj ? 0.0 : P;
      zc = g == gj ? znext : zc;
      // y_J[c'] = Σ_q Linv[q][c'] s_q on lane c' < 6, then broadcast
      double yl = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) yl += icur[q] * s[q];
      if (lane < 6 && J >= 0) yout[6 * J + lane] = yl;
      double y[6];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) y[cc] = readlane_f64(yl, cc);
      // pending sums of the earlier blocks
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) P += lcur[rr] * y[rr];
    }
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
// item_cost != nullptr: single-rank path, the reduction of the per-item [cost, invalid] pairs is done here instead
// of in a separate cost_reduce_kernel launch (with several ranks the sum goes through the all-reduce in between).
DEVI void control_prefetch(const LmState* st, const double* x_cand, int n_amb, LmState* s_st, ControlStage& pf) {
  const int tid = threadIdx.x;
  constexpr int n_int = int(sizeof(LmState) / sizeof(int));
  for (int i = tid; i < n_int; i += 256) reinterpret_cast<int*>(s_st)[i] = reinterpret_cast<const int*>(st)[i];
  pf.u0 = pf.u1 = pf.u2 = pf.u3 = 0.0; pf.c01 = nullptr;
  const double* upd = st->upd_ext;
  if (upd) {
    const int n = st->upd_ext_n;
    for (int i = tid; i < n; i += 256) { pf.u0 += upd[4 * i]; pf.u1 += upd[4 * i + 1]; pf.u2 += upd[4 * i + 2]; pf.u3 += upd[4 * i + 3]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) pf.xc[k] = (x_cand && n_amb > 0) ? x_cand[min(tid + 256 * k, n_amb - 1)] : 0.0;
}

// `st_g`: the state in global memory; the stage works on the copy `st` in LDS and writes it back at the end.
DEVI void control_body(LmState* st_g, const LmOptionsDev& o, double* R2, double* x, const double* x_cand, int n_amb, IterLog* log,
                       int log_cap, const double* __restrict__ item_cost, int n_items, const double* Rbase, size_t r_stride,
                       int no_swap, int* progress, int seq, LmState* st, const ControlStage* staged) {
  const int tid = threadIdx.x;
  ControlStage pf;
  if (staged) pf = *staged;
  else { control_prefetch(st_g, x_cand, n_amb, st, pf); }
  __syncthreads();        // the copy of the state is complete
  if (st->terminated) {
    if (progress && tid == 0) publish_progress(progress, st, seq);
    return;
  }
  __shared__ int s_accept;
  // speculative evaluation (r_stride != 0): the candidate's [cost, invalid] are the first two entries of the reduce
  // buffer the Jacobian pass at the candidate point has just filled
  if (r_stride) R2 = const_cast<double*>(Rbase + (st->rfill ? r_stride : 0));
  if (item_cost) {
    __shared__ double s_a[256], s_b[256];
    double c = 0.0, v = 0.0;
    for (int i = tid; i < n_items; i += 256) { c += item_cost[2 * i]; v += item_cost[2 * i + 1]; }
    s_a[tid] = c; s_b[tid] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) { s_a[tid] += s_a[tid + off]; s_b[tid] += s_b[tid + off]; }
      __syncthreads();
    }
    if (tid == 0) { R2[0] = s_a[0]; R2[1] = s_b[0]; }
    __syncthreads();
  }
  if (st->upd_parts == 0 && st->upd_ext) {
    // tree solver: one slot of partial sums per node, added up in slot order (fixed-shape reduction)
    // (a fixed-shape reduction: shuffle tree inside every wave, then the four waves in order)
    __shared__ double s_u[4][4];
    double u0 = pf.u0, u1 = pf.u1, u2 = pf.u2, u3 = pf.u3;
    u0 = wave_sum(u0); u1 = wave_sum(u1); u2 = wave_sum(u2); u3 = wave_sum(u3);
    if ((tid & 63) == 0) { s_u[0][tid >> 6] = u0; s_u[1][tid >> 6] = u1; s_u[2][tid >> 6] = u2; s_u[3][tid >> 6] = u3; }
    __syncthreads();
    if (tid == 0) {
      st->upd_mcc[0] = ((s_u[0][0] + s_u[0][1]) + s_u[0][2]) + s_u[0][3]; st->upd_sn[0] = ((s_u[1][0] + s_u[1][1]) + s_u[1][2]) + s_u[1][3];
      st->upd_cn[0] = ((s_u[2][0] + s_u[2][1]) + s_u[2][2]) + s_u[2][3]; st->upd_bad[0] = ((s_u[3][0] + s_u[3][1]) + s_u[3][2]) + s_u[3][3] > 0.0 ? 1 : 0;
      st->upd_parts = 1;
    }
    __syncthreads();
  }
  if (tid == 0) {
    s_accept = 0;
    finish_update(st, 0);
    const double cand_norm = st->cand_norm;
    st->n_cost_evals += 1;
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
      const double ev_cost = pf.c01 ? pf.c01[0] : R2[0], ev_invalid = pf.c01 ? pf.c01[1] : R2[1];
      const double candidate_cost = (ev_invalid > 0.0) ? 1.7976931348623157e308 : ev_cost;
      st->candidate_cost = candidate_cost;
      st->invalid_eval = ev_invalid > 0.0;
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
            st->prev_cost_change = st->last_cost_change; st->last_cost_change = st->cost_change;
            st->need_jacobian = 1;
            // the buffer evaluated at the candidate becomes R(x): by a pointer swap, or (several ranks: the host hands
            // the collective a fixed address, so the candidate is always evaluated into buffer 1) by commit_kernel
            if (r_stride) { if (no_swap) st->commit_pending = 1; else st->rcur ^= 1; }
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
  {   // the working copy back into global memory (this workgroup is the state's only writer in this launch)
    constexpr int n_int = int(sizeof(LmState) / sizeof(int));
    for (int i = tid; i < n_int; i += 256) reinterpret_cast<int*>(st_g)[i] = reinterpret_cast<const int*>(st)[i];
  }
  if (s_accept) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (tid + 256 * k < n_amb) x[tid + 256 * k] = pf.xc[k];
    for (int i = tid + 1024; i < n_amb; i += 256) x[i] = x_cand[i];
  }
  if (st->terminated) {   // (uniform)
    __syncthreads();
    publish_results_block(st_g, tid, 256);
    __syncthreads();
    if (tid == 0) st_g->published = 1;
  }
  if (progress && tid == 0) publish_progress(progress, st, seq);
}
__global__ __launch_bounds__(256) void lm_control_kernel(LmState* st, LmOptionsDev o, double* R2, double* x,
                                                         const double* x_cand, int n_amb, IterLog* log, int log_cap,
                                                         const double* __restrict__ item_cost, int n_items,
                                                         const double* Rbase, size_t r_stride, int no_swap, int* progress,
                                                         int seq) {
  __shared__ LmState s_st;
  control_body(st, o, R2, x, x_cand, n_amb, log, log_cap, item_cost, n_items, Rbase, r_stride, no_swap, progress, seq, &s_st);
}

// Several ranks, speculative evaluation: copy the accepted candidate's reduce buffer (1) over R(x) (0). The flag is
// taken down by the post_eval kernel that follows, so that the copy happens once per accepted step even when the
// iterations enqueued behind a terminated solve keep re-reducing buffer 1.
__global__ __launch_bounds__(256) void commit_kernel(const LmState* st, double* R, size_t n) {
  if (!st->commit_pending) return;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) R[i] = R[n + i];
}

// Test hook (calico_debug_lm_control_replay): the control stage driven by a given sequence of step qualities. Row i is
// seeded as "x_cost = model cost change = 1, candidate cost = 1 - rho[i]" (or a candidate that could not be evaluated),
// everything else -- radius, decrease factor, counters -- carries over from row to row as in a solve.
__global__ __launch_bounds__(256) void debug_control_replay_kernel(LmState* st, LmOptionsDev o, const double* rho, const int* infinite,
                                                                    int n, double* R2, double* radius_out, int* accepted_out,
                                                                    double* cost_out, IterLog* log, int log_cap) {
  __shared__ LmState s_st;
  for (int i = 0; i < n; ++i) {
    if (threadIdx.x == 0) {
      st->terminated = 0; st->x_cost = 1.0; st->x_norm = 1.0; st->chol_failed = 0;
      st->upd_parts = 1; st->upd_mcc[0] = 1.0; st->upd_sn[0] = 1.0; st->upd_cn[0] = 1.0; st->upd_bad[0] = 0;
      R2[0] = 1.0 - rho[i]; R2[1] = infinite[i] ? 1.0 : 0.0;
    }
    __syncthreads();
    control_body(st, o, R2, nullptr, nullptr, 0, log, log_cap, nullptr, 0, nullptr, 0, 0, nullptr, 0, &s_st);
    __syncthreads();
    if (threadIdx.x == 0) {
      radius_out[i] = st->radius; accepted_out[i] = st->step_successful;
      // accepted: the iteration's row is written after the re-evaluation at the new point (its cost is the candidate's);
      // rejected: the control stage has just written the row
      cost_out[i] = st->step_successful ? st->candidate_cost : log[st->n_log - 1].cost;
    }
    __syncthreads();
  }
}
void launch_debug_control_replay(LmState* st, const LmOptionsDev& o, const double* rho, const int* infinite, int n, double* R2,
                                 double* radius_out, int* accepted_out, double* cost_out, IterLog* log, int log_cap, hipStream_t s) {
  hipLaunchKernelGGL(debug_control_replay_kernel, dim3(1), dim3(256), 0, s, st, o, rho, infinite, n, R2, radius_out, accepted_out, cost_out,
                     log, log_cap);
}

// Results of a solve straight into host-visible pinned memory (final state, iteration log, parameter vector): a few
// kilobytes written by one workgroup over the fabric instead of three DMA copies of ~13 us each.
__global__ __launch_bounds__(256) void publish_results_kernel(const LmState* st, const IterLog* log, int log_rows, const double* x, int n_amb,
                                                              LmState* h_state, IterLog* h_log, double* h_x) {
  const int tid = threadIdx.x;
  const int n_state = int(sizeof(LmState) / sizeof(int)), n_row = int(sizeof(IterLog) / sizeof(int));
  const int rows = min(max(st->n_log, 0), log_rows);
  for (int i = tid; i < n_state; i += 256) reinterpret_cast<int*>(h_state)[i] = reinterpret_cast<const int*>(st)[i];
  for (int i = tid; i < rows * n_row; i += 256) reinterpret_cast<int*>(h_log)[i] = reinterpret_cast<const int*>(log)[i];
  for (int i = tid; i < n_amb; i += 256) h_x[i] = x[i];
}
void launch_publish_results(const LmState* st, const IterLog* log, int log_rows, const double* x, int n_amb, LmState* h_state,
                            IterLog* h_log, double* h_x, hipStream_t s) {
  hipLaunchKernelGGL(publish_results_kernel, dim3(1), dim3(256), 0, s, st, log, log_rows, x, n_amb, h_state, h_log, h_x);
}
// The parameter vector of a solve comes in the same way: read from the pinned staging buffer by the kernel that resets
// the LM state (one launch, no DMA copy).
__global__ __launch_bounds__(256) void seed_x_kernel(double* x, const double* h_x, int n_amb) {
  for (int i = threadIdx.x + blockIdx.x * 256; i < n_amb; i += 256 * gridDim.x) x[i] = h_x[i];
}
void launch_seed_x(double* x, const double* h_x, int n_amb, hipStream_t s) {
  hipLaunchKernelGGL(seed_x_kernel, dim3(4), dim3(256), 0, s, x, h_x, n_amb);
}

// Start of a solve in one launch: the parameter vector from the pinned staging buffer into x (and into the candidate
// buffer when a constant block may have changed: the update stage never writes those) and the LM state reset.
__global__ __launch_bounds__(256) void begin_solve_kernel(LmState* st, double radius, double x_norm, const double* upd_ext, int upd_ext_n,
                                                          ResultSink sink, double* x, double* x_cand, const double* h_x, int n_amb) {
  for (int i = threadIdx.x + blockIdx.x * 256; i < n_amb; i += 256 * gridDim.x) {
    const double v = h_x[i];
    x[i] = v;
    if (x_cand) x_cand[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    LmState s = {};
    s.upd_ext = upd_ext; s.upd_ext_n = upd_ext_n; s.upd_parts = 1;
    s.radius = radius; s.decrease_factor = 2.0; s.x_norm = x_norm; s.need_jacobian = 1; s.rfill = 1;
    s.min_cost = 1.7976931348623157e308;
    s.sink = sink;
    *st = s;
  }
}
void launch_begin_solve(LmState* st, double radius, double x_norm, const double* upd_ext, int upd_ext_n, const ResultSink& sink, double* x,
                        double* x_cand, const double* h_x, int n_amb, hipStream_t s) {
  hipLaunchKernelGGL(begin_solve_kernel, dim3(4), dim3(256), 0, s, st, radius, x_norm, upd_ext, upd_ext_n, sink, x, x_cand, h_x, n_amb);
}

__global__ void init_state_kernel(LmState* st, double radius, double x_norm, const double* upd_ext, int upd_ext_n) {
  LmState s = {};
  s.upd_ext = upd_ext; s.upd_ext_n = upd_ext_n; s.upd_parts = 1;
  s.radius = radius; s.decrease_factor = 2.0; s.x_norm = x_norm; s.need_jacobian = 1; s.rfill = 1;
  s.min_cost = 1.7976931348623157e308;
  *st = s;
}

// ---- launch helpers ---------------------------------------------------------
void launch_gather(double* R, const double* src, const int* out_idx_thin, const int64_t* ptr_thin, const int* idx_thin,
                   int n_thin, int n_thin8, int n_thin4, int thin_per_lane, const int* out_idx_fat, const int64_t* ptr_fat, const int* idx_fat, int n_fat,
                   const double* cost_src, int n_cost,
                   const LmState* st, int need_flag, size_t other_stride, hipStream_t s, const ControlTail* tail) {
  const int nb_fat = (n_fat + 3) / 4;
  const int nb8 = (n_thin8 + 31) / 32, nb4 = (n_thin4 - n_thin8 + 63) / 64, nb1 = (n_thin - n_thin4 + 255) / 256;
  // (A/B switch, read per launch: CALICO_GATHER_XCD=0 = the thin workgroups in output order, rounds 1-4)
  const int xcd_map = [] { const char* e = std::getenv("CALICO_GATHER_XCD"); return !e || std::atoi(e) != 0; }() ? 1 : 0;
  const int n_blocks = xcd_map ? ((1 + nb_fat + 7) & ~7) + ((nb8 + 7) & ~7) + ((nb4 + 7) & ~7) + ((nb1 + 7) & ~7) : 1 + nb_fat + nb8 + nb4 + nb1;
  ControlTail t;
  if (tail) t = *tail; else { t = ControlTail(); t.enabled = 0; }
  // workgroup 0: cost / invalid count (+ control stage), then the fat outputs, then the thin ones
  // ptr_thin == nullptr: idx_thin is the fixed-stride table (gather_pack_fixed)
#define LAUNCH_GATHER(UU, FX) hipLaunchKernelGGL((gather_kernel<UU, FX>), dim3(n_blocks), dim3(256), 0, s, R, src, out_idx_thin, ptr_thin, idx_thin, n_thin, n_thin8, n_thin4, \
                       out_idx_fat, ptr_fat, idx_fat, n_fat, nb_fat, cost_src, n_cost, st, need_flag, other_stride, t, xcd_map)
  if (thin_per_lane <= 6) { if (ptr_thin) LAUNCH_GATHER(6, false); else LAUNCH_GATHER(6, true); }
  else { if (ptr_thin) LAUNCH_GATHER(12, false); else LAUNCH_GATHER(12, true); }
#undef LAUNCH_GATHER
}
// The thin outputs' CSR lists repacked at a fixed stride per lane class (see gather_kernel, FIXED); once per plan.
__global__ __launch_bounds__(256) void gather_pack_fixed_kernel(const int64_t* __restrict__ ptr, const int* __restrict__ idx, int n_thin, int n_thin8,
                                                               int n_thin4, int s8, int zero_slot, int* __restrict__ out) {
  const int o = int(blockIdx.x) * 256 + int(threadIdx.x);
  if (o >= n_thin) return;
  const size_t base4 = size_t(n_thin8) * s8, base1 = base4 + size_t(n_thin4 - n_thin8) * 24;
  const int stride = o < n_thin8 ? s8 : (o < n_thin4 ? 24 : 8);
  const size_t base = o < n_thin8 ? size_t(o) * s8 : (o < n_thin4 ? base4 + size_t(o - n_thin8) * 24 : base1 + size_t(o - n_thin4) * 8);
  const int64_t q0 = ptr[o], q1 = ptr[o + 1];
  for (int k = 0; k < stride; ++k) out[base + k] = q0 + k < q1 ? idx[q0 + k] : zero_slot;
}
size_t gather_fixed_entries(int n_thin, int n_thin8, int n_thin4, int thin_per_lane) {
  return size_t(n_thin8) * (8 * thin_per_lane) + size_t(n_thin4 - n_thin8) * 24 + size_t(n_thin - n_thin4) * 8;
}
void launch_gather_pack_fixed(const int64_t* ptr, const int* idx, int n_thin, int n_thin8, int n_thin4, int thin_per_lane, int zero_slot, int* out,
                              hipStream_t s) {
  if (n_thin <= 0) return;
  hipLaunchKernelGGL(gather_pack_fixed_kernel, dim3((n_thin + 255) / 256), dim3(256), 0, s, ptr, idx, n_thin, n_thin8, n_thin4, 8 * thin_per_lane, zero_slot, out);
}
void launch_post_eval(const SolveArgs& a, const double* x, const BlockDev* blocks, int n_blocks, const LmOptionsDev& o,
                      IterLog* log, int log_cap, int first, int jacobi, hipStream_t s) {
  hipLaunchKernelGGL(post_eval_kernel, dim3(1), dim3(256), 0, s, a, x, blocks, n_blocks, o, log, log_cap, first, jacobi);
}
constexpr int kBorderSlice = 16;   // border columns per workgroup of the banded factorisation
size_t band_cholesky_lds_bytes(const SolveArgs& a) {
  return (size_t(a.k + 4) * (a.W() * 6 + kBorderSlice * 6 + 16 + 40) + 224) * sizeof(double);
}
size_t reduced_solve_lds_bytes(const SolveArgs& a) {
  const int m1 = a.m + 1;
  return (size_t(m1) * (m1 | 1) + m1) * sizeof(double);
}
size_t band_backsolve_lds_bytes(const SolveArgs&) { return 0; }
hipError_t configure_solve_kernels(size_t band_lds, size_t reduced_lds, size_t back_lds) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&band_cholesky_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, int(band_lds));
  if (e != hipSuccess) return e;
  if (reduced_lds) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&reduced_solve_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, int(reduced_lds));
    if (e != hipSuccess) return e;
  }
  const int big = 150 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&reduced_solve_panel_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&reduced_solve_panel_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
  for (const void* f : {reinterpret_cast<const void*>(&reduced_solve_reg_kernel<4, true>), reinterpret_cast<const void*>(&reduced_solve_reg_kernel<7, true>),
                        reinterpret_cast<const void*>(&reduced_solve_reg_kernel<10, true>), reinterpret_cast<const void*>(&reduced_solve_reg_kernel<13, true>)})
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  for (const void* f : {reinterpret_cast<const void*>(&band_backsolve_kernel<2>), reinterpret_cast<const void*>(&band_backsolve_kernel<3>),
                        reinterpret_cast<const void*>(&band_backsolve_kernel<4>), reinterpret_cast<const void*>(&band_backsolve_kernel<5>),
                        reinterpret_cast<const void*>(&band_backsolve_kernel<6>), reinterpret_cast<const void*>(&band_backsolve_kernel<7>),
                        reinterpret_cast<const void*>(&band_backsolve_kernel<8>)}) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(back_lds));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
static int reduced_blocked_from() {
  static const int v = [] { const char* e = std::getenv("CALICO_REDUCED_BLOCKED_FROM"); return e ? std::atoi(e) : 129; }();
  return v;
}
static bool reduced_is_blocked(const SolveArgs& a) {
  const int m1 = a.m + 1;
  return m1 >= reduced_blocked_from() && m1 > 128 && a.m <= 256 * kRBCols;
}
// K-slices of the Schur complement the reduced solve adds up on load
int reduced_schur_slices(const SolveArgs& a) {
  if (!(a.m + 1 <= 128 || reduced_is_blocked(a))) return 1;
  // Long trajectories: the Schur complement's tiles walk all eliminated rows (6 n_cp · 32/30), a few tiles x K-slices
  // workgroups in all -- at 1453 control points two slices meant 4.7k rows per workgroup, 12 us beside the last level.
  // More slices there; the in-LDS 32-column-block solver adds up to eight on load (the 16-column panel solver two).
  const char* e = std::getenv("CALICO_DENSE");
  const bool panel = e && std::string(e) == "panel";
  // (CALICO_SCHUR_SLICES=1 / 2: A/B switch, read per solve)
  if (const char* se = std::getenv("CALICO_SCHUR_SLICES")) return std::max(1, std::min(kSchurSlices, std::atoi(se)));
  if (a.m + 1 <= 128 && !panel) {
    if (a.n_cp >= 1280) return 8;
    if (a.n_cp >= 640) return 4;
    // Short trajectories: ONE slice. The tiles' workgroups ride beside the last level's chains and have time to spare
    // (< 1k eliminated rows each), while every slice is 16 more load instructions per thread at the head of the reduced
    // solve, the launch's critical path (configs[3]: 30.6 -> 29.9 us, level 1 unchanged).
    if (a.n_cp < 160) return 1;
  }
  // (the blocked factorisation of a larger system adds its slices up in its first step, the longest of its launches:
  //  configs[4], 222 control points: 3425 -> 3548 it/s with one slice; at 440 control points the tiles' workgroups would
  //  end the last level's launch -- there the in-LDS path loses 2.9 % with one)
  if (a.n_cp < 320) return 1;
  return kSchurSlices;
}
// Dense solve of the (m+1)x(m+1) augmented reduced system in a.Spart (ks K-slices) -> a.y[n_s ...]
void launch_dense_block_solve(const SolveArgs& a, int ks, hipStream_t s, int t0 = 0, int outer_back = 0);     // bcr_kernels.hip
void launch_reduced_block_step(const SolveArgs& a, int j, int nsl, int n_wg, hipStream_t s);     // bcr_kernels.hip
bool launch_reduced_fused(const SolveArgs& a, int nsteps, int nsl, int* words, hipStream_t s);     // bcr_kernels.hip
void launch_reduced_solve(const SolveArgs& a, bool reduced_in_lds, int ks, hipStream_t s, int* fan_words) {
  const int m1 = a.m + 1;
  const bool blocked = reduced_is_blocked(a);
  // CALICO_DENSE=panel keeps the 16-column panel kernel for the in-LDS case (A/B switch)
  const bool use_block = [] { const char* e = std::getenv("CALICO_DENSE"); return !(e && std::string(e) == "panel"); }();      // (read per solve: A/B switches)
  if (m1 <= 128 && a.m >= 1 && use_block && ks <= 8) { launch_dense_block_solve(a, ks, s); return; }
  if (m1 <= 128) {
    const size_t lds = (size_t(m1) * ((16 * ((m1 + 15) / 16)) | 1) + m1 + 32 + 128 + 256) * sizeof(double);
    if (m1 <= 64) hipLaunchKernelGGL(reduced_solve_panel_kernel<1>, dim3(1), dim3(256), lds, s, a, 0, ks);
    else hipLaunchKernelGGL(reduced_solve_panel_kernel<2>, dim3(1), dim3(256), lds, s, a, 0, ks);
  } else if (blocked) {
    // panels are eliminated over all CUs until what is left fits the in-LDS solver, which finishes the factorisation
    // and solves for its unknowns; the blocked backward sweep takes it from there
    const int steps = (m1 - 128 + kRB - 1) / kRB, t0 = kRB * steps, mt1 = m1 - t0;
    // CALICO_BLOCK_STEP=valu: round 2's step kernel (in-wave column Cholesky on 64 rows, VALU tile update) and backward
    // sweep in a launch of its own -- A/B switch
    const bool step_mfma = [] { const char* e = std::getenv("CALICO_BLOCK_STEP"); return !(e && std::string(e) == "valu"); }();
    const bool fused_back = step_mfma && use_block && mt1 >= 2 && a.m <= 1024;     // (the dense solver goes on with the panels' backward sweep)
    // CALICO_REDUCED_FUSED=1: all of it in ONE launch (reduced_fused_kernel: the steps behind fan-ins of one another, the
    // in-LDS solver last). Off by default: a step has nothing to do before the step in front of it is through, so a fan-in
    // only replaces the boundary -- by L1-bypassing loads and write-through stores of everything that crosses it: configs[4]
    // 3545 -> 3520 it/s (read per solve).
    const bool one_launch = fused_back && fan_words && [] { const char* e = std::getenv("CALICO_REDUCED_FUSED"); return e && std::atoi(e) != 0; }();
    if (one_launch && launch_reduced_fused(a, steps, ks, fan_words, s)) return;
    for (int j = 0; j < steps; ++j) {
      const int rows = m1 - kRB * (j + 1), T = rows > 0 ? (rows + 63) / 64 : 0;
      if (step_mfma) launch_reduced_block_step(a, j, j == 0 ? ks : 1, T > 0 ? T * (T + 1) / 2 : 1, s);
      else hipLaunchKernelGGL(reduced_block_step_kernel, dim3(T > 0 ? T * (T + 1) / 2 : 1), dim3(256), 0, s, a, j, j == 0 ? ks : 1);
    }
    if (use_block && mt1 >= 2) launch_dense_block_solve(a, 1, s, t0, fused_back ? 1 : 0);     // (the 32-column-block solver of the small systems)
    else {
      const size_t lds = (size_t(mt1) * ((16 * ((mt1 + 15) / 16)) | 1) + mt1 + 32 + 128 + 256) * sizeof(double);
      if (mt1 <= 64) hipLaunchKernelGGL(reduced_solve_panel_kernel<1>, dim3(1), dim3(256), lds, s, a, t0, 1);
      else hipLaunchKernelGGL(reduced_solve_panel_kernel<2>, dim3(1), dim3(256), lds, s, a, t0, 1);
    }
    if (!fused_back) hipLaunchKernelGGL(reduced_block_back_kernel, dim3(1), dim3(256), 0, s, a, steps);
  } else if (m1 <= 16 * 13) {
    const int NT = m1 <= 64 ? 4 : (m1 <= 112 ? 7 : (m1 <= 160 ? 10 : 13));
    const int NP = 16 * NT;
    const size_t full = size_t(3 * NP + size_t(a.m + 1) * NP) * sizeof(double);
    const bool l_in_lds = full <= 150 * 1024;
    const size_t lds = l_in_lds ? full : size_t(3 * NP) * sizeof(double);
#define LAUNCH_RS(N) \
    if (l_in_lds) hipLaunchKernelGGL((reduced_solve_reg_kernel<N, true>), dim3(1), dim3(256), lds, s, a); \
    else hipLaunchKernelGGL((reduced_solve_reg_kernel<N, false>), dim3(1), dim3(256), lds, s, a)
    switch (NT) {
      case 4: LAUNCH_RS(4); break;
      case 7: LAUNCH_RS(7); break;
      case 10: LAUNCH_RS(10); break;
      default: LAUNCH_RS(13); break;
    }
#undef LAUNCH_RS
  } else {
    hipLaunchKernelGGL(reduced_solve_kernel, dim3(1), dim3(256), reduced_in_lds ? reduced_solve_lds_bytes(a) : 0, s, a,
                       reduced_in_lds ? 1 : 0);
  }
}
void launch_solve(const SolveArgs& a, const LmOptionsDev& o, const double* x, double* x_cand, const BlockDev* blocks,
                  int n_blocks, bool reduced_in_lds, hipStream_t s, bool with_post_eval, IterLog* log, int log_cap, int jacobi) {
  const int m1 = a.m + 1;
  const size_t total = size_t(a.n_cp) * a.W() * 6 + size_t(a.n_s()) * m1 + size_t(m1) * m1;
  const int pb = int((total + 255) / 256);
  hipLaunchKernelGGL(prepare_kernel, dim3((pb < 2048 ? pb : 2048) + (with_post_eval ? 1 : 0)), dim3(256), 0, s, a, o, with_post_eval ? 1 : 0,
                     x, blocks, n_blocks, log, log_cap, jacobi);
  const int nwg = (m1 + kBorderSlice - 1) / kBorderSlice;
  hipLaunchKernelGGL(band_cholesky_kernel, dim3(nwg, a.n_seg()), dim3(256), band_cholesky_lds_bytes(a), s, a, kBorderSlice);
  const int nt = (m1 + 15) / 16;
  const int ks = reduced_schur_slices(a);
  hipLaunchKernelGGL(schur_kernel, dim3(nt * (nt + 1) / 2 * ks), dim3(256), 0, s, a, ks);
  launch_reduced_solve(a, reduced_in_lds, ks, s, nullptr);
  hipLaunchKernelGGL(border_matvec_kernel, dim3((a.n_s() + 3) / 4), dim3(256), 0, s, a);
  {
    const size_t bl = band_backsolve_lds_bytes(a);
    switch (a.k) {
      case 2: hipLaunchKernelGGL(band_backsolve_kernel<2>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
      case 3: hipLaunchKernelGGL(band_backsolve_kernel<3>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
      case 4: hipLaunchKernelGGL(band_backsolve_kernel<4>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
      case 5: hipLaunchKernelGGL(band_backsolve_kernel<5>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
      case 6: hipLaunchKernelGGL(band_backsolve_kernel<6>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
      case 7: hipLaunchKernelGGL(band_backsolve_kernel<7>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
      default: hipLaunchKernelGGL(band_backsolve_kernel<8>, dim3(a.n_seg()), dim3(256), bl, s, a, x, x_cand, blocks, n_blocks); break;
    }
  }
}
void launch_cost_reduce(const double* item_cost, int n_items, double* R2, const LmState* st, hipStream_t s) {
  hipLaunchKernelGGL(cost_reduce_kernel, dim3(1), dim3(256), 0, s, item_cost, n_items, R2, st);
}
void launch_control(LmState* st, const LmOptionsDev& o, double* R2, double* x, const double* x_cand, int n_amb,
                    IterLog* log, int log_cap, const double* item_cost, int n_items, const double* Rbase, size_t r_stride,
                    hipStream_t s, bool commit_by_copy, int* progress, int seq) {
  hipLaunchKernelGGL(lm_control_kernel, dim3(1), dim3(256), 0, s, st, o, R2, x, x_cand, n_amb, log, log_cap, item_cost, n_items,
                     Rbase, r_stride, commit_by_copy ? 1 : 0, progress, seq);
  if (commit_by_copy && r_stride)
    hipLaunchKernelGGL(commit_kernel, dim3(unsigned(std::min<size_t>(512, (r_stride + 255) / 256))), dim3(256), 0, s, st,
                       const_cast<double*>(Rbase), r_stride);
}
void launch_init_state(LmState* st, double radius, double x_norm, hipStream_t s, const double* upd_ext, int upd_ext_n) {
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, s, st, radius, x_norm, upd_ext, upd_ext_n);
}

}  // namespace cal
