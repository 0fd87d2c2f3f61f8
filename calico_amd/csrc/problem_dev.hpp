// problem_dev.hpp — device-side problem description shared by host and kernels.
//
// Data layout in HBM (all FP64 unless noted):
//   x[n_amb]            ambient parameter values of every block (current point),
//   x_cand[n_amb]       candidate point of the LM step,
//   obs_* SoA arrays    observations sorted by (layout, segment), i.e. by "cell":
//                       every residual block of a cell touches the same k control
//                       points and the same calibration blocks, so its Jacobian
//                       rows share one local column set,
//   partials            per work item: (c+1)×(c+1) [JᵀJ | Jᵀr] block + cost + flag,
//   R (reduce buffer)   [cost | invalid | g(NT) | band blocks | border E | corner C],
//                       the only thing exchanged between GPUs.
#pragma once
#include <stddef.h>
#include <stdint.h>

// Development instrumentation of the kernels (cycle counts printed by one wave per kernel, life span of every wave of the
// Jacobian launch: CALICO_KERNEL_TIMING). Compiled in only with -DCALICO_DEV_TIMING (CALICO_DEV_TIMING=1 in the
// environment of __graft_entry__.build()): switched off at run time it still cost ~4 us per LM iteration in registers
// and branches on the latency chains.
#ifdef CALICO_DEV_TIMING
#define CAL_DEV_TIMING(expr) (expr)
#else
#define CAL_DEV_TIMING(expr) false
#endif

namespace cal {

constexpr int kRowsPerItem = 128;   // LDS rows staged per work item (64 camera obs × 2)
constexpr int kRowPad = 129;        // largest row stride (doubles) of a staged Jacobian column
// eval_cells_kernel: the dynamic LDS its launch may ask for. ONE constant for the plan's decision (cell workgroups or
// not), the kernel's attribute and the launch: cells_launch_lds_bytes() is what a launch asks for.
constexpr size_t kCellsMaxLds = 158 * 1024;
constexpr size_t cells_launch_lds_bytes(size_t wave_lds_doubles) { return (2 * wave_lds_doubles + 2) * sizeof(double); }

struct SensorDev {
  int kind, model, K, loss;
  int intr_off, q_off, t_off, lat_off, grav_off;  // ambient offsets into x
  int pad0;
  double info, loss_scale;
};

// Local column layout of one (sensor, rigid body) pair. Column order:
// [spline 6k | intrinsics | q | t | latency | body q | body t | gravity | model point];
// an entry is -1 when that block is constant (no column).
struct LayoutDev {
  int sensor, ncols;
  int c_intr, c_q, c_t, c_lat, c_bq, c_bt, c_grav;
  int bq_off, bt_off;  // ambient offsets of the rigid body pose (cameras)
  int c_pt;            // free model point (cameras): the layout is then per (sensor, body, point)
};

// One work item = up to kRowsPerItem residual rows of one cell.
struct ItemDev {
  int layout, seg, obs_begin, obs_count;
  int64_t partial_off;  // offset (doubles) of this item's partial block
  int64_t rows_off;     // >= 0: the item files its staged rows [J r] here instead (lds_cols × row_pad doubles, column-major)
                        // and the cell kernel forms [J r]ᵀ[J r] of the whole cell; -1: the item forms its own block
  // copies of what the wave would otherwise reach through item -> layout -> sensor and item -> segment -> control points:
  // every dependent global load costs about a microsecond at the head of a latency chain
  LayoutDev L;
  SensorDev S;
  int ctrl_off[8];      // ambient offsets of the segment's control points (spline order <= 8)
};

// A camera frame: residual blocks of one cell that also share the time stamp, hence the pose,
// its time derivative and the spline weights. Their Jacobian rows factor as J = J_prim · T_frame.
struct FrameItemDev {
  int layout, seg, obs_begin, obs_count;
  double stamp;
  int64_t partial_off;   // offset (doubles) of the frame's COMPACT record: M_ext (PE×PE) then coef (ncols+1)
  LayoutDev L;           // copies, as in ItemDev
  SensorDev S;
  int ctrl_off[8];
  // the frame's cell (cell workgroups, EvalArgs.pair_mode): copies of the cell descriptor's fields (no load of it in front of
  // the expansion)
  int cell, cell_frames, cell_prim_off, cell_pad;
  int64_t cell_partial_off, cell_src_off;
};

// All frames of one cell = (layout, segment): the cell kernel expands and sums their compact records
// into one (c+1)×(c+1) partial block, which is what the gather sees.
// A cell of IMU items (prim_off = -1) reuses the descriptor: frame_begin / frame_count count its work items, src_off is
// the row store of the first one (contiguous, lds_cols × row_pad doubles each), PE the rows of a full item and pad0 the
// rows of the whole cell.
struct CellDev {
  int layout, seg, frame_begin, frame_count;
  int64_t partial_off;   // expanded (c+1)×(c+1) block of the cell
  int64_t src_off;       // first compact record (the cell's records are contiguous)
  int n1, PE;            // c+1, side of M_ext
  int prim_off, pad0;    // offset into the per-layout prim-column table
};

struct EvalArgs {
  const double* x;
  const SensorDev* sensors;
  const LayoutDev* layouts;
  const ItemDev* items;
  const double* knots;
  const double* basis;    // per segment k×k
  const int* ctrl_off;    // ambient offset of every control point
  const double* m0; const double* m1; const double* m2; const double* stamp;
  const int* point_off;   // ambient offset of the observed model point (cameras)
  double* partials;
  double* item_cost;      // per item: [cost, invalid]
  double* res_out;        // residual write-back (n_obs × 3), or nullptr
  uint8_t* valid_out;
  int order, n_items, lds_cols, apply_loss;
  const struct LmState* st;  // optional: skip when terminated (and, with need_flag, when no Jacobian is due)
  int need_flag, cost_index_base;
  const FrameItemDev* fitems;
  int n_fitems, debug;   // debug: CALICO_KERNEL_TIMING cycle print-outs
  int row_pad, n_cells;  // row_pad: row stride (doubles) of a staged Jacobian column: max rows per item + 1, odd
  const CellDev* cells;
  const int* prim_tab;   // per frame layout: prim column (row/col of M_ext) of every local column
  int cell_chunk, cell_rec_max;   // frames per LDS chunk of the cell kernel, largest compact record (doubles)
  int project, row_cell_chunk;   // prediction mode of the cost-only kernel (measurements read as 0, 1/sigma as -1 -> output =
                                 // model); work items per LDS pass of a row cell
  const uint8_t* active; // per observation (sorted order): 0 = tagged as outlier, left out; nullptr = all in
  int frame_lds_doubles, pad5;   // LDS of a frame workgroup for the widest frame layout of the problem (0: worst case)
  unsigned long long* wave_log;  // CALICO_KERNEL_TIMING=3: [start, end] of every workgroup of the Jacobian launch (100 MHz clock)
  // Streaming solve loop: one more workgroup of the Jacobian launch looks at what the linear solve in front of it left
  // (model cost change, step norm) and tells the host whether the control stage of this iteration is going to end the solve;
  // the host enqueues the next iteration only on "go" (progress word 2), so that a solve that ends leaves no iteration of
  // early-exit kernels behind on the stream. nullptr: no hint.
  int* hint_progress;
  int hint_seq, hint_first;       // hint_first: the hint's workgroup is block 0 of eval_cells_kernel (else the last one)
  double hint_ftol, hint_ptol;
  // Cell workgroups (plans with `fuse_expand`; eval_cells_kernel): the Jacobian launch runs workgroups of TWO waves -- the
  // (at most two) frames of one camera cell, which then expand the cell's block together out of LDS: no compact record
  // leaves the CU and expand_cells_kernel has no launch --, or two work items that form their blocks themselves.
  int pair_mode;             // a.fitems holds two entries per camera cell (the second one may be empty: obs_count = 0)
  int wave_lds_doubles;      // LDS of one wave of such a workgroup
};

// Where the stage that terminates a solve leaves its results for the host (pinned, host-mapped memory: final state,
// iteration log, parameter vector), and the solve's number: the progress words the host polls carry it, so that the
// early-exit kernels a terminated solve leaves behind on the stream cannot be mistaken for the next solve's.
struct IterLog;
struct LmState;
struct ResultSink {
  LmState* state; IterLog* log; double* x;   // host side (nullptr: results are fetched by publish_results_kernel)
  const IterLog* src_log; const double* src_x;
  int rows, n_amb, epoch, pad;
};

// LM state kept on the device; the control kernel is its only writer.
struct LmState {
  double radius, decrease_factor;
  double x_cost, candidate_cost, model_cost_change;
  double x_norm, cand_norm, step_norm, gradient_max_norm, gradient_norm;
  double relative_decrease, cost_change;
  double prev_cost_change;  // cost change of the successful step before the last one (the host sizes its batches with the ratio)
  double last_cost_change;  // cost change of the last successful step
  double initial_cost, min_cost;
  int iteration;
  int need_jacobian;      // the last step was accepted: re-evaluate J at x
  int terminated, termination_type, termination_reason;
  int step_valid, step_successful, chol_failed;
  int num_consecutive_invalid, num_successful, num_unsuccessful;
  int invalid_eval;       // candidate evaluation hit an invalid projection
  int n_log, n_jac_evals, n_cost_evals;
  int rcur;               // which of the two reduce buffers holds R(x) (speculative evaluation)
  int rfill;              // the reduce buffer the speculative evaluation of this iteration fills (= rcur ^ 1, latched by the update
                          // stage: the control stage may flip rcur while workgroups of the gather are still starting)
  // partial sums of the update stage, one slot per band segment (the control stage adds them up in slot order)
  double upd_mcc[2], upd_sn[2], upd_cn[2];
  int upd_bad[2], upd_parts;
  int commit_pending;     // multi-rank speculative evaluation: the accepted candidate's buffer 1 is to be copied over buffer 0
  // tree solver (bcr_kernels.hip): every node of the elimination tree leaves the partial sums of the update stage
  // [model cost change, |step|^2, |candidate|^2, non-finite flag] in its own slot; the control stage adds them up in
  // slot order (upd_parts == 0 selects this form)
  const double* upd_ext;
  int upd_ext_n;
  int last_logged_iteration;   // iteration number of the last row of the log, kept even when the log buffer is full
  int published;               // the results went to `sink` (once per solve)
  ResultSink sink;
};

struct LmOptionsDev {
  int max_num_iterations, max_num_consecutive_invalid_steps;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
};

struct IterLog {  // mirrors calico_iteration
  int iteration, step_is_valid, step_is_successful, reserved;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
};

struct BlockDev {  // one reduced (free, used) parameter block
  int amb_off, size, manifold, tan_off;
};

// LM control stage riding in the gather kernel (single rank, speculative evaluation): the workgroup that sums the
// candidate's [cost, invalid] (outputs 0 and 1 of the gather, always in one workgroup) takes the accept / reject
// decision at once, while the other workgroups are still assembling the normal equations.
// What the device needs to build the source lists of the band, the border and the spline right-hand side itself
// (solve_kernels.hip, launch_gather_lists). `tab` (ints): the expanded block of every cell [n_lay][nseg] (offset into the
// partials, -1: no such cell), the local column of every calibration tangent column in every layout [n_lay][m] (-1: not a
// column of that layout), the block side n1 of every layout [n_lay].
struct GatherStruct {
  const int* tab;
  int n_lay, nseg, n_cp, k, m;      // m: calibration tangent columns (SolveArgs.mc)
  int d_split;                       // band blocks at distance < d_split from the diagonal are numbered first (struct_output)
  unsigned long long off_g, off_B, off_E;
};

struct ControlTail {
  int enabled, n_amb, log_cap, seq;
  LmOptionsDev o;
  double* x;
  const double* x_cand;
  IterLog* log;
  const double* Rbase;
  size_t r_stride;
  int* progress;
  int owner_block;   // workgroup that produces outputs 0 and 1
};

#if defined(__HIPCC__)
#define CAL_HD __host__ __device__
#else
#define CAL_HD
#endif
// A cell's / work item's partial block [J r]^T[J r] is its UPPER TRIANGLE, packed row-major (round 5; a full n1 x n1 square
// with only the upper triangle written before): entry (i, j), i <= j, at tri_off(i, j, n1); tri_size(n1) entries per block.
// The gather touched every other 128-byte line of the squares half-used -- at configs[4] the blocks of an XCD's stretch of
// the trajectory (59 MB over all XCDs as squares) no longer fitted its L2.
CAL_HD inline int tri_off(int i, int j, int n1) { return i * n1 - ((i * (i - 1)) >> 1) + (j - i); }
CAL_HD inline int tri_size(int n1) { return (n1 * (n1 + 1)) >> 1; }

// Arguments of the linear-solve kernels. The reduce buffer R is laid out as
// [cost | invalid | g(NT) | band blocks B(n_cp,k,6,6) | border E(6n_cp,m) | corner C(m,m)].
struct SolveArgs {
  const double* R;        // reduce buffer 0; buffer 1 follows at r_stride doubles (0: single-buffered)
  size_t r_stride;
  double* Lb;             // [n_cp][6k][6] band factor by block column: Lb[J][r][c] = L(6J+r, 6J+c)
  double* Linv;           // [n_cp][6][6]  inverse of every 6x6 pivot block
  double* Y;              // [n_s][m+1] L^-1 [E | g_s]
  double* S;              // [m+1][m+1] C + damping; row/column m carry g_c
  double* Spart;          // [m+1][m+1] reduced system S - YtY (lower triangle)
  double* Swork;          // global fallback workspace of the reduced solve
  double* y;              // [NT] solution of the damped system (unscaled): delta = -y
  double* zbuf;           // [n_s] z = L^-1 g_s - Y y_c (input of the backward band sweep)
  double* dadd;           // [NT] damping added to the diagonal
  double* scale;          // [NT] Jacobi scaling 1/(1+sqrt(H_jj)) from iteration 0
  const uint8_t* cp_active;  // [n_cp]
  LmState* st;
  int n_cp, k;
  int m;                  // width of the dense border the solver kernels work with = mc + 6·sep_n
  int mc;                 // tangent size of the free calibration blocks (border width of R)
  // Nested dissection with one separator: control points [sep_s, sep_s + sep_n), sep_n = k-1 or 0, are taken out
  // of the band and appended to the border (columns mc..m-1). The band then falls apart into two independent
  // segments [0, sep_s) and [sep_s + sep_n, n_cp) whose sequential sweeps run side by side.
  int sep_s, sep_n;
  int debug;              // CALICO_KERNEL_TIMING=1: kernels print per-phase cycle counts (development aid)
  // host-mapped progress words, or nullptr: [0] = sequence number of the last LM iteration the control kernel has
  // finished with, [1] = LmState.terminated. The host polls them instead of synchronising (single-rank solve loop).
  int* progress;
  CAL_HD int n_s() const { return 6 * n_cp; }
  CAL_HD int W() const { return 6 * k; }
  CAL_HD int NT() const { return 6 * n_cp + mc; }
  CAL_HD size_t off_g() const { return 2; }
  CAL_HD size_t off_B() const { return 2 + size_t(NT()); }
  CAL_HD size_t off_E() const { return off_B() + size_t(n_cp) * k * 36; }
  CAL_HD size_t off_C() const { return off_E() + size_t(n_s()) * mc; }
  CAL_HD size_t r_size() const { return off_C() + size_t(mc) * mc; }
  CAL_HD bool in_sep(int tangent_row) const { return sep_n > 0 && tangent_row >= 6 * sep_s && tangent_row < 6 * (sep_s + sep_n); }
  CAL_HD int seg_begin(int seg) const { return seg == 0 ? 0 : sep_s + sep_n; }
  CAL_HD int seg_end(int seg) const { return sep_n > 0 && seg == 0 ? sep_s : n_cp; }
  CAL_HD int n_seg() const { return sep_n > 0 ? 2 : 1; }
  // tangent index of border index b: calibration first, then the separator rows
  CAL_HD int border_tangent(int b) const { return b < mc ? n_s() + b : 6 * sep_s + (b - mc); }
  // where the solution of tangent index j sits in y: the separator part is solved with the border
  CAL_HD int y_index(int j) const { return in_sep(j) ? n_s() + mc + (j - 6 * sep_s) : j; }
};

// ---------------------------------------------------------------------------
// Tree solver ("BCR": block cyclic reduction / nested dissection of the band, bcr_kernels.hip).
// The control points are grouped into superblocks of kBcrCps = 5 (30 rows, padded to 32): with spline order k <= 6 a
// superblock only couples to its two neighbours, so the band is block tridiagonal with a dense border (calibration
// columns + right-hand side). Level 0 eliminates chains of q consecutive superblocks between kept separators, every
// further level every other surviving separator; the last survivor (the root) joins the calibration blocks in the
// dense reduced system. The dependent chain is O(q + log(n_cp)) block factorisations instead of n_cp / 2 band steps.
// ---------------------------------------------------------------------------
constexpr int kBcrCps = 5;        // control points per superblock
constexpr int kBcrBP = 32;        // padded superblock size
constexpr int kBcrMaxChain = 8;   // longest chain a node eliminates
constexpr int kBcrFS = 16;        // border columns per workgroup of a node

struct BcrNodeDev {
  int q;                     // chain length: superblocks blk0, blk0 + 1, ..., eliminated left to right (q > 1 only at level 0)
  int left, right;           // separator superblocks on either side (-1: none)
  int slot;                  // the node's slot of update-stage partial sums
  int blk0;                  // first superblock of the chain
  int pend;                  // bit 0 / 1: the (single) superblock carries a pending update from the chain on its left / right (previous level)
};

// Node descriptors that reach a level's launch without a load (bcr_level_kernel): q_regular > 0 -- level 0, whose nodes are
// regular ([chain of q] [separator] ...), described by the chain length alone; n > 0 -- a level of at most four nodes, by value.
struct BcrInlineNodes {
  int q_regular, n;
  BcrNodeDev nd[4];
};

// The top level's nodes (single superblocks between the root and the ends) when their back-substitution rides in the
// launch of the level below: every node there solves the top separators next to it itself (n = 0: separate launch).
struct BcrTopSeps {
  int n;
  int blk[2], left[2], right[2];     // superblock, and its own separators (the root or -1)
};

struct BcrArgs {
  double* D;                 // [N][32][32]   diagonal superblocks (damped), symmetric, full storage
  double* G;                 // [2][N][32][32] coupling to the NEXT surviving superblock: G[b][r][c] = H(next row r, b column c); ping-pong by level
  double* F;                 // [N][32][m1p]  border rows: calibration columns, right-hand side in column mc, zero padding
  double* pendD;             // [2][N][2][32][32]   updates a chain leaves for its separators (to be ADDED), by level parity and side
  double* pendF;             // [2][N][2][32][m1p]
  double* M;                 // [N][32][32]   L^-T of every eliminated superblock (upper triangular)
  double* ZA;                // [N][32][32]   L^-1 · coupling to the left separator
  double* ZB;                // [N][32][32]   L^-1 · coupling to the next superblock of the chain / the right separator
  double* Y;                 // [N][32][m1p]  L^-1 · border rows (column mc: L^-1 g); rows of the root stay zero
  double* ysol;              // [N][32]       solution by superblock
  double* zb;                // [N][32]       L^-1 g - Z^F y_c by superblock row (formed once per solve for the levels below the top)
  double* upd;               // [n_slots][4]  update-stage partial sums
  const BcrNodeDev* nodes;   // all levels, level after level
  const int* keep;           // kept superblocks with pending updates to apply, level after level: [blk, mask] pairs
  const int* cp_block;       // [n_cp] BlockDev index of a control point (-1: unobserved)
  const int* ctrl_off;       // [n_cp] ambient offset of a control point (6 values, no manifold)
  int all_active, pad0;      // every control point is observed (no activity look-ups)
  int N, m1p;                // superblocks; padded border width (multiple of 16, >= mc + 1)
  int root, root_pend, root_par;   // surviving superblock (-1: none), its pending mask and the parity of those slots
  int n_slots;
};

// Words of the LM state read at the head of a kernel, by VECTOR loads. A wave-uniform address normally becomes a scalar
// load, and scalar loads return out of order: the first later use of ANY scalar load's result -- a kernel argument the
// compiler fetches late, a descriptor -- is an s_waitcnt lgkmcnt(0), i.e. it waits for the state's round trip as well,
// and every request behind it is issued a round trip late (~2k clocks behind a kernel boundary; ISA of the tree levels).
// Vector loads return in order: a load issued first is waited for with vmcnt(N), whatever was requested behind it.
// `vector_ptr` hides the pointer's uniformity from the compiler; `uniform` brings a loaded word back into an SGPR.
template <class T> __device__ __forceinline__ const T* vector_ptr(const T* p) { asm volatile("" : "+v"(p)); return p; }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Sums across lanes without the LDS crossbar. `__shfl_xor` compiles to ds_bpermute_b32 (two per double, an LDS round
// trip per step of a dependent chain); inside a row of 16 lanes the DPP modifiers exchange lanes in two v_mov_b32_dpp,
// and v_permlane32_swap / v_permlane16_swap (gfx950) exchange the halves of a wave and the rows of a half
// (`profiles/microbench/lane_sums.hip`: 132 against 336 clocks for a dependent 16-lane sum, 228 against 496 for a wave).
// Every lane receives the sum, added in a fixed order. ALL lanes of the wave must be active where these are called: a
// DPP move from a disabled lane leaves the destination at zero instead of faulting.
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row4_sum(double v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); return v; }       // quad_perm [1,0,3,2], [2,3,0,1]
__device__ __forceinline__ double row8_sum(double v) { v = row4_sum(v); v += dpp_mov<0x141>(v); return v; }              // row_half_mirror
__device__ __forceinline__ double row16_sum(double v) { v = row8_sum(v); v += dpp_mov<0x140>(v); return v; }             // row_mirror
__device__ __forceinline__ double other_half(double v) {     // lane i receives lane i ^ 32
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const bool up = (threadIdx.x & 32) != 0;
  return __hiloint2double(up ? b[0] : b[1], up ? a[0] : a[1]);
}
__device__ __forceinline__ double other_row(double v) {      // lane i receives lane i ^ 16
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const bool odd = (threadIdx.x & 16) != 0;
  return __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
}
__device__ __forceinline__ double wave_sum(double v) { v = row16_sum(v); v += other_half(v); v += other_row(v); return v; }

}  // namespace cal
