// eval_kernels.hip — residual / Jacobian / normal-equation partials on gfx950.
//
// One wave (64 lanes) per work item: a run of residual blocks that share one
// cell = (sensor, rigid body, spline segment). Stage A: one residual block per
// lane — the lane evaluates the reference's cost functor and its analytic
// Jacobian in FP64 registers (wave-uniform parameter blocks come in through
// scalar loads), applies the robust loss, and stages its Jacobian rows
// transposed in LDS (column-major, padded stride, conflict-free 16-byte
// writes). Stage B: the wave forms the item's dense JᵀJ | Jᵀr block from LDS
// with 4×4 register tiles and writes it once, coalesced. The per-observation
// Jacobian never touches HBM.
//
// Reference arithmetic reproduced here:
//   camera_cost_functor.h:71-147, gyroscope_cost_functor.h:58-118,
//   accelerometer_cost_functor.h:62-147, bspline.hpp:39-72, geometry.h:137-222.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "device_math.hpp"
#include "problem_dev.hpp"

namespace cal {

// a value every lane holds alike, handed to the compiler as such (scalar registers instead of a vector pair)
__device__ __forceinline__ double wave_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}


// Wave-uniform context of a work item.
struct ItemCtx {
  const SensorDev* s;
  const LayoutDev* L;
  int k;
  double knot0, knot1;
  const double* M;        // k×k basis of the segment
  const int* ctrl_off;    // ambient offsets of the segment's k control points
  const double* x;
  double info;            // 1/sigma of the sensor; -1 in prediction mode (measurement 0): the "residual" is the model output
};

// Evaluate spline value / derivatives at t: out[d][6] = sum_i W[d][i] * ctrl_i.
template <int ND, int KT>
DEV void spline_eval(const ItemCtx& c, double t, double W[ND][kMaxOrder], double out[ND][6]) {
  const int k = KT > 0 ? KT : c.k;
  spline_weights<ND, KT>(c.k, c.knot0, c.knot1, c.M, t, W);
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int a = 0; a < 6; ++a) out[d][a] = 0.0;
#pragma unroll
  for (int i = 0; i < kMaxOrder; ++i) {
    if (i >= k) continue;
    const double* cp = c.x + c.ctrl_off[i];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double v = cp[a];
#pragma unroll
      for (int d = 0; d < ND; ++d) out[d][a] += W[d][i] * v;
    }
  }
}

struct RowSink {
  double* J;  // LDS, column-major [col][pad]
  int row0, pad;
  DEV void put(int col, int r, double v) const { J[col * pad + row0 + r] = v; }
};

// ---------------------------------------------------------------------------
// Camera: residual (2) and Jacobian rows. Returns false on invalid projection.
// ---------------------------------------------------------------------------
template <int MODEL, bool JAC, int KT>
DEV bool camera_block(const ItemCtx& c, double px, double py, double stamp, const double* xm, double res[2],
                      const RowSink& sink, double* cost, int apply_loss) {
  const SensorDev& S = *c.s;
  const LayoutDev& L = *c.L;
  const double* intr = c.x + S.intr_off;
  const double* qp = c.x + S.q_off;
  const double* tp = c.x + S.t_off;
  const double lat = c.x[S.lat_off];
  const double* bq = c.x + L.bq_off;
  const double* bt = c.x + L.bt_off;
  Q4 q_rc; q_rc.x = qp[0]; q_rc.y = qp[1]; q_rc.z = qp[2]; q_rc.w = qp[3];
  Q4 q_wm; q_wm.x = bq[0]; q_wm.y = bq[1]; q_wm.z = bq[2]; q_wm.w = bq[3];
  const M3 R_rc = rotmat(normalized(q_rc));
  const M3 R_wm = rotmat(normalized(q_wm));
  const V3 t_rc = mk(tp[0], tp[1], tp[2]);
  const V3 t_wm = mk(bt[0], bt[1], bt[2]);
  constexpr int ND = JAC ? 2 : 1;
  double W[ND][kMaxOrder], P[ND][6];
  spline_eval<ND, KT>(c, stamp - lat, W, P);
  const V3 phi = mk(-P[0][0], -P[0][1], -P[0][2]);
  const V3 t_wr = mk(P[0][3], P[0][4], P[0][5]);
  const M3 R_rw = rotmat(angle_axis_to_quat(phi));
  const V3 Rx = mul(R_wm, mk(xm[0], xm[1], xm[2]));
  const V3 v = (Rx + t_wm) - t_wr;
  const V3 y = mul(R_rw, v);
  const V3 z = y - t_rc;
  const V3 xc = mulT(R_rc, z);
  double pix[2], D[2][3], dK[2][kMaxIntr];
  if (!project<MODEL, JAC>(intr, xc, pix, D, dK)) return false;
  double r0 = (px - pix[0]) * c.info, r1 = (py - pix[1]) * c.info;
  const double sq = r0 * r0 + r1 * r1;
  double ls = 1.0, rho = sq;
  if (apply_loss) rho = loss_eval(S.loss, S.loss_scale, sq, &ls);
  *cost = 0.5 * rho;
  res[0] = r0 * ls; res[1] = r1 * ls;
  if constexpr (JAC) {
    const double fac = -c.info * ls;
    // DRt = D·R_rcᵀ, DG = DRt·R_rw  (2×3)
    double DRt[2][3], DG[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < 3; ++j) DRt[r][j] = D[r][0] * R_rc.m[j][0] + D[r][1] * R_rc.m[j][1] + D[r][2] * R_rc.m[j][2];
#pragma unroll
      for (int j = 0; j < 3; ++j) DG[r][j] = DRt[r][0] * R_rw.m[0][j] + DRt[r][1] * R_rw.m[1][j] + DRt[r][2] * R_rw.m[2][j];
    }
    // A = dr/dp (2×6): rotation part fac·DRt·[y]×·J_l(phi); translation part -fac·DG
    const Rodrigues<double> rod = rodrigues<double>(phi.x, phi.y, phi.z, false);
    const M3 Jl = rod_J_matrix(rod);
    const M3 Sy = skew(y);
    double A[2][6];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double T1[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) T1[j] = DRt[r][0] * Sy.m[0][j] + DRt[r][1] * Sy.m[1][j] + DRt[r][2] * Sy.m[2][j];
#pragma unroll
      for (int j = 0; j < 3; ++j) A[r][j] = fac * (T1[0] * Jl.m[0][j] + T1[1] * Jl.m[1][j] + T1[2] * Jl.m[2][j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) A[r][3 + j] = -fac * DG[r][j];
    }
    // spline columns: w_i · A
#pragma unroll
    for (int i = 0; i < kMaxOrder; ++i) {
      if (i >= (KT > 0 ? KT : c.k)) continue;
      const double w = W[0][i];
#pragma unroll
      for (int a = 0; a < 6; ++a) { sink.put(6 * i + a, 0, w * A[0][a]); sink.put(6 * i + a, 1, w * A[1][a]); }
    }
    if (L.c_intr >= 0) {
      constexpr int K = CamK<MODEL>::K;
#pragma unroll
      for (int j = 0; j < K; ++j) { sink.put(L.c_intr + j, 0, fac * dK[0][j]); sink.put(L.c_intr + j, 1, fac * dK[1][j]); }
    }
    if (L.c_q >= 0) {  // d xc / d delta = 2 R_rcᵀ [z]×
      const M3 Sz = skew(z);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          sink.put(L.c_q + j, r, 2.0 * fac * (DRt[r][0] * Sz.m[0][j] + DRt[r][1] * Sz.m[1][j] + DRt[r][2] * Sz.m[2][j]));
    }
    if (L.c_t >= 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) sink.put(L.c_t + j, r, -fac * DRt[r][j]);
    }
    if (L.c_lat >= 0) {  // dp/dlat = -pdot
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) s += A[r][a] * P[ND - 1][a];
        sink.put(L.c_lat, r, -s);
      }
    }
    if (L.c_bq >= 0) {  // d xc / d delta_wm = -2 G [R_wm x_m]×
      const M3 Sx = skew(Rx);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          sink.put(L.c_bq + j, r, -2.0 * fac * (DG[r][0] * Sx.m[0][j] + DG[r][1] * Sx.m[1][j] + DG[r][2] * Sx.m[2][j]));
    }
    if (L.c_bt >= 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) sink.put(L.c_bt + j, r, fac * DG[r][j]);
    }
    if (L.c_pt >= 0) {  // d xc / d x_m = G R_wm
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          sink.put(L.c_pt + j, r, fac * (DG[r][0] * R_wm.m[0][j] + DG[r][1] * R_wm.m[1][j] + DG[r][2] * R_wm.m[2][j]));
    }
  }
  return true;
}

template <bool JAC, int KT>
DEV bool camera_dispatch(const ItemCtx& c, double px, double py, double stamp, const double* xm, double res[2],
                         const RowSink& sink, double* cost, int apply_loss) {
  switch (c.s->model) {
    case 1: return camera_block<1, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
    case 2: return camera_block<2, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
    case 3: return camera_block<3, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
    case 4: return camera_block<4, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
    case 5: return camera_block<5, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
    case 6: return camera_block<6, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
    default: return camera_block<7, JAC, KT>(c, px, py, stamp, xm, res, sink, cost, apply_loss);
  }
}

// ---------------------------------------------------------------------------
// IMU kinematics shared by the gyroscope and accelerometer blocks.
// ---------------------------------------------------------------------------
// The Jacobian rows of an IMU block are formed by THREE lanes (`jl` = 0, 1, 2 = lane of the observation's triple): all
// of them evaluate the residual, lane jl differentiates along φ_jl and files the columns of every 3-wide group that
// belong to component jl -- the single-lane chain these blocks were is the longest of the Jacobian launch.
// Element [q][jl] of a 3×3 matrix every lane of the triple holds (pick3: no control flow on jl, see device_math.hpp)
DEV double col_of(const M3& A, int q, const Lane3& l) { return pick3(l, A.m[q][0], A.m[q][1], A.m[q][2]); }
DEV double col_of(const double A[3][3], int q, const Lane3& l) { return pick3(l, A[q][0], A[q][1], A[q][2]); }
DEV double pick(const double v[6], int off, const Lane3& l) { return pick3(l, v[off], v[off + 1], v[off + 2]); }
// sum over the triple, in lane order (the latency column: Σ over the three pose components), valid in lane 0 of the triple
DEV double triple_sum(double t, int jl) {
  const int base = int(threadIdx.x & 63) - jl;
  const double t1 = __shfl(t, base + 1, 64), t2 = __shfl(t, base + 2, 64);
  return (t + t1) + t2;
}

// omega = J(phi)·phid and column jl of W = d omega / d phi (dual number along φ_jl).
DEV void omega_and_dphi_col(V3 phi, V3 phid, const Lane3& l3, V3* omega, double Wcol[3], Rodrigues<double>* Rv) {
  D1 px = mk1(phi.x), py = mk1(phi.y), pz = mk1(phi.z);
  px.d = pick3(l3, 1.0, 0.0, 0.0); py.d = pick3(l3, 0.0, 1.0, 0.0); pz.d = pick3(l3, 0.0, 0.0, 1.0);
  const Rodrigues<D1> R = rodrigues<D1>(px, py, pz, false);
  *Rv = rod_value(R);
  D1 ox, oy, oz;
  rod_J_apply<D1>(R, mk1(phid.x), mk1(phid.y), mk1(phid.z), &ox, &oy, &oz);
  *omega = mk(ox.v, oy.v, oz.v);
  Wcol[0] = ox.d; Wcol[1] = oy.d; Wcol[2] = oz.d;
}

template <bool JAC, int KT>
DEV bool gyro_block(const ItemCtx& c, V3 meas, double stamp, double res[3], const RowSink& sink, double* cost,
                    int apply_loss, int jl) {
  const SensorDev& S = *c.s;
  const LayoutDev& L = *c.L;
  const double* intr = c.x + S.intr_off;
  const double* qp = c.x + S.q_off;
  const double lat = c.x[S.lat_off];
  Q4 q; q.x = qp[0]; q.y = qp[1]; q.z = qp[2]; q.w = qp[3];
  const M3 R_rg = rotmat(normalized(q));
  constexpr int ND = JAC ? 3 : 2;
  double W[ND][kMaxOrder], P[ND][6];
  spline_eval<ND, KT>(c, stamp - lat, W, P);
  const V3 phi = mk(-P[0][0], -P[0][1], -P[0][2]);
  const V3 phid = mk(-P[1][0], -P[1][1], -P[1][2]);
  V3 omega;
  double Wcol[3] = {0.0, 0.0, 0.0};
  M3 Jl;
  const Lane3 l3 = lane3(jl);
  if constexpr (JAC) {
    Rodrigues<double> Rv;
    omega_and_dphi_col(phi, phid, l3, &omega, Wcol, &Rv);
    Jl = rod_J_matrix(Rv);
  } else {
    const Rodrigues<double> R = rodrigues<double>(phi.x, phi.y, phi.z, false);
    rod_J_apply<double>(R, phid.x, phid.y, phid.z, &omega.x, &omega.y, &omega.z);
  }
  const V3 og = -mulT(R_rg, omega);
  double f[3], Mw[3][3], dK[3][kMaxIntr];
  imu_project<JAC>(S.model, intr, og, f, Mw, dK);
  double r[3] = {(meas.x - f[0]) * c.info, (meas.y - f[1]) * c.info, (meas.z - f[2]) * c.info};
  const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double ls = 1.0, rho = sq;
  if (apply_loss) rho = loss_eval(S.loss, S.loss_scale, sq, &ls);
  *cost = 0.5 * rho;
#pragma unroll
  for (int i = 0; i < 3; ++i) res[i] = r[i] * ls;
  if constexpr (JAC) {
    const double fac = -c.info * ls;
    // B = fac · Mw · (-R_rgᵀ)  : dr/d omega_rw
    double B[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) B[i][j] = -fac * (Mw[i][0] * R_rg.m[j][0] + Mw[i][1] * R_rg.m[j][1] + Mw[i][2] * R_rg.m[j][2]);
    // column jl of A0 = dr/dp_r = B·W·(-1) and of A1 = dr/dpdot_r = B·J·(-1)
    double A0c[3], A1c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      A0c[i] = -(B[i][0] * Wcol[0] + B[i][1] * Wcol[1] + B[i][2] * Wcol[2]);
      A1c[i] = -(B[i][0] * col_of(Jl, 0, l3) + B[i][1] * col_of(Jl, 1, l3) + B[i][2] * col_of(Jl, 2, l3));
    }
#pragma unroll
    for (int i = 0; i < kMaxOrder; ++i) {
      if (i >= (KT > 0 ? KT : c.k)) continue;
      const double w0 = W[0][i], w1 = W[1][i];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        sink.put(6 * i + jl, rr, w0 * A0c[rr] + w1 * A1c[rr]);
        sink.put(6 * i + 3 + jl, rr, 0.0);
      }
    }
    if (L.c_intr >= 0) {
      // lane jl files the columns j = jl, jl + 3, ... (its component of every 3-wide group): the group's three candidates are
      // picked without control flow, only the bound on K predicates the stores
      const int K = imu_num_params(S.model);
#pragma unroll
      for (int g3 = 0; g3 < kMaxIntr; g3 += 3) {
        if (g3 >= K) continue;      // (wave-uniform)
        const int j = g3 + jl;
        double v[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) v[rr] = fac * pick3(l3, dK[rr][g3], dK[rr][g3 + 1], dK[rr][g3 + 2]);
        if (j < K) {
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) sink.put(L.c_intr + j, rr, v[rr]);
        }
      }
    }
    if (L.c_q >= 0) {  // d og / d delta = -2 R_rgᵀ [omega]×
      const M3 So = skew(omega);
      double RtSc[3];   // column jl of R_rgᵀ [omega]×
#pragma unroll
      for (int i = 0; i < 3; ++i) RtSc[i] = R_rg.m[0][i] * col_of(So, 0, l3) + R_rg.m[1][i] * col_of(So, 1, l3) + R_rg.m[2][i] * col_of(So, 2, l3);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
        sink.put(L.c_q + jl, rr, -2.0 * fac * (Mw[rr][0] * RtSc[0] + Mw[rr][1] * RtSc[1] + Mw[rr][2] * RtSc[2]));
    }
    if (L.c_t >= 0) {  // the translation extrinsic has zero derivative (gyroscope_cost_functor.h:94-114)
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) sink.put(L.c_t + jl, rr, 0.0);
    }
    if (L.c_lat >= 0) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const double s = triple_sum(A0c[rr] * pick(P[1], 0, l3) + A1c[rr] * pick(P[ND - 1], 0, l3), jl);
        if (jl == 0) sink.put(L.c_lat, rr, -s);
      }
    }
  }
  return true;
}

template <bool JAC, int KT>
DEV bool accel_block(const ItemCtx& c, V3 meas, double stamp, double res[3], const RowSink& sink, double* cost,
                     int apply_loss, int jl) {
  const SensorDev& S = *c.s;
  const LayoutDev& L = *c.L;
  const double* intr = c.x + S.intr_off;
  const double* qp = c.x + S.q_off;
  const double* tp = c.x + S.t_off;
  const double* gp = c.x + S.grav_off;
  const double lat = c.x[S.lat_off];
  Q4 q; q.x = qp[0]; q.y = qp[1]; q.z = qp[2]; q.w = qp[3];
  const M3 R_ra = rotmat(normalized(q));
  const V3 t = mk(tp[0], tp[1], tp[2]);
  const V3 g = mk(gp[0], gp[1], gp[2]);
  constexpr int ND = JAC ? 4 : 3;
  double W[ND][kMaxOrder], P[ND][6];
  spline_eval<ND, KT>(c, stamp - lat, W, P);
  const V3 phi = mk(-P[0][0], -P[0][1], -P[0][2]);
  const V3 phid = mk(-P[1][0], -P[1][1], -P[1][2]);
  const V3 phidd = mk(-P[2][0], -P[2][1], -P[2][2]);
  const V3 aw = mk(P[2][3], P[2][4], P[2][5]);
  const M3 R_rw = rotmat(angle_axis_to_quat(phi));
  V3 omega, alpha;
  double dw_col[3] = {0.0, 0.0, 0.0}, da_col[3] = {0.0, 0.0, 0.0};   // column jl of d omega / d phi, d alpha / d phi
  double Hphid_row[3] = {0.0, 0.0, 0.0};                              // row jl of H(phi)·phid
  Rodrigues<double> Rd;   // the coefficients at phi as plain doubles (Jacobian path)
  Rd.zero = true;
  const Lane3 l3 = lane3(jl);
  const double e0 = pick3(l3, 1.0, 0.0, 0.0), e1 = pick3(l3, 0.0, 1.0, 0.0), e2 = pick3(l3, 0.0, 0.0, 1.0);     // unit vector of this lane's direction
  if constexpr (JAC) {
    D1 px = mk1(phi.x), py = mk1(phi.y), pz = mk1(phi.z);
    px.d = e0; py.d = e1; pz.d = e2;
    const Rodrigues<D1> R = rodrigues<D1>(px, py, pz, true);
    Rd = rod_value(R);
    D1 o[3], jdd[3], Hv[3][3];
    rod_J_apply<D1>(R, mk1(phid.x), mk1(phid.y), mk1(phid.z), &o[0], &o[1], &o[2]);
    rod_J_apply<D1>(R, mk1(phidd.x), mk1(phidd.y), mk1(phidd.z), &jdd[0], &jdd[1], &jdd[2]);
    rod_H_apply<D1>(R, mk1(phid.x), mk1(phid.y), mk1(phid.z), Hv);
    D1 al[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) al[j] = phid.x * Hv[0][j] + phid.y * Hv[1][j] + phid.z * Hv[2][j] + jdd[j];
    omega = mk(o[0].v, o[1].v, o[2].v);
    alpha = mk(al[0].v, al[1].v, al[2].v);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dw_col[j] = o[j].d; da_col[j] = al[j].d;
      Hphid_row[j] = pick3(l3, Hv[0][j].v, Hv[1][j].v, Hv[2][j].v);
    }
  } else {
    const Rodrigues<double> R = rodrigues<double>(phi.x, phi.y, phi.z, true);
    double jdd[3], Hv[3][3];
    rod_J_apply<double>(R, phid.x, phid.y, phid.z, &omega.x, &omega.y, &omega.z);
    rod_J_apply<double>(R, phidd.x, phidd.y, phidd.z, &jdd[0], &jdd[1], &jdd[2]);
    rod_H_apply<double>(R, phid.x, phid.y, phid.z, Hv);
    alpha = mk(phid.x * Hv[0][0] + phid.y * Hv[1][0] + phid.z * Hv[2][0] + jdd[0],
               phid.x * Hv[0][1] + phid.y * Hv[1][1] + phid.z * Hv[2][1] + jdd[1],
               phid.x * Hv[0][2] + phid.y * Hv[1][2] + phid.z * Hv[2][2] + jdd[2]);
  }
  // b = R_rw (a_w - g) + omega×(omega×t) - alpha×t ;  f = R_raᵀ b
  const V3 rag = mul(R_rw, aw - g);
  const V3 b = rag + cross(omega, cross(omega, t)) - cross(alpha, t);
  const V3 fs = mulT(R_ra, b);
  double f[3], Mw[3][3], dK[3][kMaxIntr];
  imu_project<JAC>(S.model, intr, fs, f, Mw, dK);
  double r[3] = {(meas.x - f[0]) * c.info, (meas.y - f[1]) * c.info, (meas.z - f[2]) * c.info};
  const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  double ls = 1.0, rho = sq;
  if (apply_loss) rho = loss_eval(S.loss, S.loss_scale, sq, &ls);
  *cost = 0.5 * rho;
#pragma unroll
  for (int i = 0; i < 3; ++i) res[i] = r[i] * ls;
  if constexpr (JAC) {
    const double fac = -c.info * ls;
    // Bm = fac · Mw · R_raᵀ : dr/db
    double Bm[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Bm[i][j] = fac * (Mw[i][0] * R_ra.m[j][0] + Mw[i][1] * R_ra.m[j][1] + Mw[i][2] * R_ra.m[j][2]);
    // db/d omega = (omega·t) I + omega tᵀ - 2 t omegaᵀ ;  db/d alpha = [t]×
    const double ot = dot(omega, t);
    double dbdw[3][3];
    const M3 St = skew(t);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        dbdw[i][j] = (i == j ? ot : 0.0) + comp(omega, i) * comp(t, j) - 2.0 * comp(t, i) * comp(omega, j);
    // column jl of d alpha / d phid: with H[i][j][l] = (H_i e_l)_j it is Σ_i (H[i][j][jl] + H[jl][j][i]) phid_i; the first
    // sum needs the slice of this lane's direction, the second is row jl of H·phid (already there)
    const M3 Jl = rod_J_matrix(Rd);
    double Hs[3][3];
    rod_H_apply<double>(Rd, e0, e1, e2, Hs);
    double dad_col[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) dad_col[j] = (Hs[0][j] * phid.x + Hs[1][j] * phid.y + Hs[2][j] * phid.z) + Hphid_row[j];
    // column jl of db/dphi = -[rag]× J_l + dbdw·dw_dphi + St·da_dphi, of db/dphid and of db/dphidd
    const M3 Sr = skew(rag);
    double dbc[3], dbdc[3], dbddc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < 3; ++q2) {
        const double jq = col_of(Jl, q2, l3);
        s0 += -Sr.m[i][q2] * jq + dbdw[i][q2] * dw_col[q2] + St.m[i][q2] * da_col[q2];
        s1 += dbdw[i][q2] * jq + St.m[i][q2] * dad_col[q2];
        s2 += St.m[i][q2] * jq;
      }
      dbc[i] = s0; dbdc[i] = s1; dbddc[i] = s2;
    }
    // column jl of A0 = dr/dp_r, A1 = dr/dpdot_r, A2r = dr/dpddot_r (phi = -p_r), A2t = dr/dpddot_t = Bm·R_rw
    double A0c[3], A1c[3], A2rc[3], A2tc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < 3; ++q2) {
        s0 += Bm[i][q2] * dbc[q2]; s1 += Bm[i][q2] * dbdc[q2];
        s2 += Bm[i][q2] * dbddc[q2]; s3 += Bm[i][q2] * col_of(R_rw, q2, l3);
      }
      A0c[i] = -s0; A1c[i] = -s1; A2rc[i] = -s2; A2tc[i] = s3;
    }
#pragma unroll
    for (int i = 0; i < kMaxOrder; ++i) {
      if (i >= (KT > 0 ? KT : c.k)) continue;
      const double w0 = W[0][i], w1 = W[1][i], w2 = W[2][i];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        sink.put(6 * i + jl, rr, w0 * A0c[rr] + w1 * A1c[rr] + w2 * A2rc[rr]);
        sink.put(6 * i + 3 + jl, rr, w2 * A2tc[rr]);
      }
    }
    if (L.c_intr >= 0) {
      // lane jl files the columns j = jl, jl + 3, ... (its component of every 3-wide group): the group's three candidates are
      // picked without control flow, only the bound on K predicates the stores
      const int K = imu_num_params(S.model);
#pragma unroll
      for (int g3 = 0; g3 < kMaxIntr; g3 += 3) {
        if (g3 >= K) continue;      // (wave-uniform)
        const int j = g3 + jl;
        double v[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) v[rr] = fac * pick3(l3, dK[rr][g3], dK[rr][g3 + 1], dK[rr][g3 + 2]);
        if (j < K) {
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) sink.put(L.c_intr + j, rr, v[rr]);
        }
      }
    }
    if (L.c_q >= 0) {  // d f / d delta = 2 R_raᵀ [b]×
      const M3 Sb = skew(b);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
        sink.put(L.c_q + jl, rr, 2.0 * (Bm[rr][0] * col_of(Sb, 0, l3) + Bm[rr][1] * col_of(Sb, 1, l3) + Bm[rr][2] * col_of(Sb, 2, l3)));
    }
    if (L.c_t >= 0) {  // db/dt = [omega]×[omega]× - [alpha]×
      const M3 So = skew(omega), Sa = skew(alpha);
      const M3 So2 = mul(So, So);
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        double s = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < 3; ++q2) s += Bm[rr][q2] * (col_of(So2, q2, l3) - col_of(Sa, q2, l3));
        sink.put(L.c_t + jl, rr, s);
      }
    }
    if (L.c_lat >= 0) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const double t = A0c[rr] * pick(P[1], 0, l3) + A1c[rr] * pick(P[2], 0, l3) + A2rc[rr] * pick(P[ND - 1], 0, l3) +
                         A2tc[rr] * pick(P[ND - 1], 3, l3);
        const double s = triple_sum(t, jl);
        if (jl == 0) sink.put(L.c_lat, rr, -s);
      }
    }
    if (L.c_grav >= 0) {  // db/dg = -R_rw
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) sink.put(L.c_grav + jl, rr, -A2tc[rr]);
    }
  }
  return true;
}



typedef double f64x4 __attribute__((ext_vector_type(4)));
// [J r]ᵀ[J r] of the staged rows (column-major, stride pad) -> upper triangle of the n1×n1 item block, NT = ceil(n1/16).
// T0 / T1: tiles [T0, T1) of the upper triangle in row-major order only -- two waves of a workgroup sharing one item's tiles
// (eval_cells_kernel); a tile's k-steps stay in one wave, in the same order.
// (The staged rows are named as LDS: through a generic pointer -- what a function that is not inlined gets -- every operand
//  was a flat_load. The file is compiled with -amdgpu-mfma-vgpr-form: see __graft_entry__.py.)
// A cell's block leaves with write-through stores: 6 MB of blocks written at the very end of the launch's workgroups would
// otherwise sit dirty in the XCDs' L2s and be written back behind the last wave, inside the launch's duration.
DEV void block_store(double* p, double v) {
#ifdef CALICO_BLOCK_STORE_PLAIN
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
template <int NT> struct StageBTiles {      // tile t of the row-major upper triangle -> (row[t], col[t]), at compile time
  int row[NT * (NT + 1) / 2], col[NT * (NT + 1) / 2];
  constexpr StageBTiles() : row(), col() {
    int k = 0;
    for (int I = 0; I < NT; ++I)
      for (int J = I; J < NT; ++J) { row[k] = I; col[k] = J; ++k; }
  }
};
typedef const double __attribute__((address_space(3))) * LdsRows;
template <int NT, int T0 = 0, int T1 = NT * (NT + 1) / 2>
// (inlined since round 5: a function that is not inlined waits for its stores to complete before it returns -- ~1.3k clocks per
//  call for the tiles' write-through stores, three calls per IMU pair; inlined, the stores of a wave's last call complete behind
//  the wave. With the staged rows named as LDS and the trip count scalar the inlined body no longer needs its own register budget.)
DEV void stage_b_mfma(LdsRows lds, int pad, int nrows, int n1, double* out) {
  const int lane = threadIdx.x & 63, lc16 = lane & 15, lk = lane >> 4;
  constexpr int NP = T1 - T0;
  constexpr StageBTiles<NT> tiles{};
  constexpr int G0 = tiles.row[T0];      // the first column group any of the tiles reads
  f64x4 acc[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  // Round 5: NO arithmetic between the products (a wave's VALU instructions and its FP64 MFMAs do not overlap:
  // profiles/microbench/mfma_f64_rate.hip). The staging area holds whole groups of sixteen columns and whole groups of four
  // rows -- the columns past n1 and the rows past nrows are ZEROS (eval_items_body clears them) --, so an operand is a plain
  // load: no clamped index, no 0/1 factor (eight v_mul_f64 and the index arithmetic per k-step before: ~89 clocks per product
  // against 64). Two k-steps per trip on two sets of operand registers: the next k-step's operands are requested in front of
  // this one's products without a register move, and the pointers advance once per trip.
  LdsRows cp[NT];
#pragma unroll
  for (int t = G0; t < NT; ++t) cp[t] = lds + (16 * t + lc16) * pad + lk;
  // (the trip count in a scalar register -- a function that is not inlined gets its arguments in vector registers, and with a
  //  trip count the compiler takes for lane-dependent the loop is exec-masked blocks between which it cannot count the
  //  outstanding LDS requests: it waited for ALL of them, the ones just issued included, in front of every group of products)
  const int ksteps = __builtin_amdgcn_readfirstlane((nrows + 3) >> 2);
  double oa[NT], ob[NT];
#pragma unroll
  for (int t = G0; t < NT; ++t) oa[t] = cp[t][0];
  int k = 0;
  for (; k + 2 < ksteps; k += 2) {        // k-steps k and k + 1; k + 2's operands requested
#pragma unroll
    for (int t = G0; t < NT; ++t) ob[t] = cp[t][4];
    __builtin_amdgcn_sched_barrier(0);      // (left to itself the scheduler sinks these requests into the next trip, right in front of their use)
#pragma unroll
    for (int t = 0; t < NP; ++t)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[tiles.row[T0 + t]], oa[tiles.col[T0 + t]], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = G0; t < NT; ++t) oa[t] = cp[t][8];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NP; ++t)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[tiles.row[T0 + t]], ob[tiles.col[T0 + t]], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = G0; t < NT; ++t) cp[t] += 8;
  }
  if (k + 1 < ksteps) {                   // two k-steps left
#pragma unroll
    for (int t = G0; t < NT; ++t) ob[t] = cp[t][4];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NP; ++t)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[tiles.row[T0 + t]], oa[tiles.col[T0 + t]], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NP; ++t)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ob[tiles.row[T0 + t]], ob[tiles.col[T0 + t]], acc[t], 0, 0, 0);
  } else if (k < ksteps) {                // one
#pragma unroll
    for (int t = 0; t < NP; ++t)
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[tiles.row[T0 + t]], oa[tiles.col[T0 + t]], acc[t], 0, 0, 0);
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) + 4·reg. Forty stores per ten tiles: the block is named as GLOBAL memory
  // with 32-bit offsets from one base (through the generic pointer every store was a flat_store behind a 64-bit multiply-add:
  // 470 instructions, ~6k clocks per call with the stores' completion -- as much as two thirds of the products), the packed
  // row's start is formed once per (tile row, register) and shared by the tiles of the row, and a tile above the diagonal
  // needs no row / column comparison.
  typedef double __attribute__((address_space(1))) * GlobalBlock;
  GlobalBlock og = (GlobalBlock)out;
  // start of packed row gi = lk + 4 s (s = 4 I + r), less gi: tri_off(gi, gj, n1) = rs[s] + gj. By recurrence -- rs(gi + 4) =
  // rs(gi) + 4 n1 - 4 gi - 10 --: two additions per row instead of two multiplications
  unsigned rs[4 * NT];
  {
    int gi = lk;
    int v = gi * n1 - ((gi * (gi - 1)) >> 1) - gi;
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) { rs[q] = unsigned(v); v += 4 * n1 - 4 * gi - 10; gi += 4; }
  }
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    const int I = tiles.row[T0 + t], J = tiles.col[T0 + t];
    const int gj = 16 * J + lc16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 16 * I + lk + 4 * r;
      const bool keep = (I < J || gi <= gj) && gj < n1;
      if (keep) {
        const unsigned off = rs[4 * I + r] + unsigned(gj);      // (unsigned: one base in scalar registers + a 32-bit offset per store)
#ifdef CALICO_BLOCK_STORE_PLAIN
        og[off] = acc[t][r];
#else
        __hip_atomic_store(og + off, acc[t][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      }
    }
  }
}
// PART 0: all tiles; 1: the first three quarters of the upper triangle's tiles in row-major order (the late wave's share: it starts on
// them the moment its rows are staged); 2: the rest (the early wave's, behind its own block).
// (Every call costs ~4k clocks on top of its MFMAs, so a share is one call: splitting the early wave's own tiles around the
//  hand-over as well measured slower.)
template <int NT, int PART>
DEV void stage_b_part(LdsRows lds, int pad, int nrows, int n1, double* out) {
  // (round 5: three quarters, it was 60 % -- with the IMU blocks' lane-dependent choices free of branches the late wave's
  //  rows are staged at 24.8k clocks instead of 39.6k, the early wave is through with its own block at 35k: at 60 % the early
  //  wave ended at 44.6k, the late one at 36.5k)
  constexpr int T = NT * (NT + 1) / 2, n_main = (3 * T + 2) / 4;
  if constexpr (PART == 0) stage_b_mfma<NT>(lds, pad, nrows, n1, out);
  else if constexpr (PART == 1) { if constexpr (n_main > 0) stage_b_mfma<NT, 0, n_main>(lds, pad, nrows, n1, out); }
  else { if constexpr (n_main < T) stage_b_mfma<NT, n_main, T>(lds, pad, nrows, n1, out); }
}
template <int PART>
DEV void stage_b_dispatch(const double* lds_generic, int pad, int nrows, int n1, double* out) {
  LdsRows lds = (LdsRows)lds_generic;
  switch ((n1 + 15) >> 4) {
    case 1: stage_b_part<1, PART>(lds, pad, nrows, n1, out); break;
    case 2: stage_b_part<2, PART>(lds, pad, nrows, n1, out); break;
    case 3: stage_b_part<3, PART>(lds, pad, nrows, n1, out); break;
    case 4: stage_b_part<4, PART>(lds, pad, nrows, n1, out); break;
    default: stage_b_part<5, PART>(lds, pad, nrows, n1, out); break;
  }
}

// LDS traffic inside ONE wave (the frame and work-item kernels run one wave per workgroup): the wave's own LDS stores
// are complete before its lanes read what the others wrote. __syncthreads() would also wait for the global loads
// requested ahead (the next batch's observations).
DEV void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}


// ---------------------------------------------------------------------------
// The evaluation kernel. JAC: stage Jacobian rows and form the item's
// [JᵀJ | Jᵀr] block; !JAC: residuals / cost only.
// grid = n_items, block = 64 (one wave).
// ---------------------------------------------------------------------------
struct ItemStage { int nrows, n1; double* out; };      // a work item whose rows are staged, [J r]ᵀ[J r] still to be formed
template <bool JAC, int KT>
DEV void eval_items_body(const EvalArgs& a, const int item_id, double* lds, ItemStage* stage_only = nullptr) {
  if (a.st && (a.st->terminated || (a.need_flag && !a.st->need_jacobian))) return;
  const int lane = threadIdx.x & 63;        // (one wave per item; a workgroup of eval_cells_kernel has two)
  const int row_pad = a.row_pad;
  const bool dbg = CAL_DEV_TIMING(JAC && a.debug && (item_id == 3 || item_id == a.n_items - 2) && lane == 0);
  long long tph[4] = {0, 0, 0, 0}, tk = dbg ? __builtin_readcyclecounter() : 0;
#define ITICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  const ItemDev* ip = a.items + item_id;
  const ItemDev it = *ip;
  const LayoutDev& L = ip->L;       // (copies inside the item record: one hop instead of item -> layout -> sensor)
  const SensorDev& S = ip->S;
  ItemCtx c;
  c.s = &S; c.L = &L; c.k = a.order; c.x = a.x; c.info = wave_uniform(a.project ? -1.0 : S.info);      // (in SGPRs: as a VGPR pair it was the one value the work items spilled to scratch, reloaded sixteen times)
  const int ki = it.seg + a.order - 1;
  c.knot0 = a.knots[ki]; c.knot1 = a.knots[ki + 1];
  c.M = a.basis + size_t(it.seg) * a.order * a.order;
  c.ctrl_off = ip->ctrl_off;
  const int dim = (S.kind == 0) ? 2 : 3;
  // The IMU blocks are the longest single-lane chains of the launch (accelerometer ≈ 45k clocks): where such a wave
  // shares a SIMD with a camera wave it gets the issue slots first, the launch ends when the last of them does.
  if (JAC) { if (S.kind == 2) __builtin_amdgcn_s_setprio(3); else if (S.kind == 1) __builtin_amdgcn_s_setprio(2); }
  const int ncols = L.ncols;           // Jacobian columns; column ncols holds the residual
  const int cp4 = (ncols + 1 + 3) & ~3;
  const int nrows = dim * it.obs_count;
  // (no zeroing of the staging area: every active lane writes all columns of its rows, and stage B masks the
  //  rows / columns beyond the item)
  ITICK(0)
  // IMU Jacobian rows: three lanes per observation (see gyro_block); everything else: one lane per observation
  const bool triple = JAC && S.kind != 0;
  const int ol = triple ? lane / 3 : lane, jl = triple ? lane - 3 * ol : 0;
  const bool active = ol < it.obs_count;
  const int o = it.obs_begin + ol;
  // observations tagged as outliers (camera.cpp:121-124: skipped by AddResidualsToProblem) stay in the arrays but
  // contribute nothing: zero rows, zero cost, not an evaluation failure
  const bool on = active && (!a.active || a.active[o] != 0);
  double res[3] = {0.0, 0.0, 0.0};
  double cost = 0.0;
  bool ok = true;
  RowSink sink; sink.J = lds; sink.row0 = dim * ol; sink.pad = row_pad;
  if (on) {
    const double st = a.stamp[o];
    const double z0 = a.project ? 0.0 : a.m0[o], z1 = a.project ? 0.0 : a.m1[o], z2 = a.project ? 0.0 : a.m2[o];
    if (S.kind == 0) {
      ok = camera_dispatch<JAC, KT>(c, z0, z1, st, a.x + a.point_off[o], res, sink, &cost, a.apply_loss);
    } else if (S.kind == 1) {
      ok = gyro_block<JAC, KT>(c, mk(z0, z1, z2), st, res, sink, &cost, a.apply_loss, jl);
    } else {
      ok = accel_block<JAC, KT>(c, mk(z0, z1, z2), st, res, sink, &cost, a.apply_loss, jl);
    }
    if (!ok) { cost = 0.0; res[0] = res[1] = res[2] = 0.0; }
  }
  if (active && jl == 0 && a.res_out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < dim) a.res_out[size_t(o) * 3 + r] = res[r];
    a.valid_out[o] = (on && ok) ? 1 : 0;
  }
  ITICK(1)
  const double item_cost = wave_sum(jl == 0 ? cost : 0.0);
  const double n_invalid = wave_sum((on && !ok && jl == 0) ? 1.0 : 0.0);
  if (lane == 0) { a.item_cost[2 * (a.cost_index_base + item_id)] = item_cost; a.item_cost[2 * (a.cost_index_base + item_id) + 1] = n_invalid; }
  if constexpr (JAC) {
    if (active && jl == 0) {
      if (!ok || !on) {  // drop the block: zero its rows
        for (int col = 0; col < ncols; ++col)
          for (int r = 0; r < dim; ++r) sink.put(col, r, 0.0);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) if (r < dim) sink.put(ncols, r, res[r]);
    }
    const int n1 = ncols + 1;
    if (it.rows_off < 0) {
      // stage B reads whole groups of sixteen columns and of four rows without masks: the columns [n1, 16 ceil(n1 / 16)) and
      // the rows [nrows, 4 ceil(nrows / 4)) of the staging area are zeros (the host sizes the area for them: lds_cols, row_pad)
      const int nrows4 = (nrows + 3) & ~3, ncols16 = (n1 + 15) & ~15;
      const int extra_r = nrows4 - nrows;
      for (int e = lane; e < extra_r * n1; e += 64) lds[(e / extra_r) * row_pad + nrows + e % extra_r] = 0.0;
      for (int e = lane; e < (ncols16 - n1) * nrows4; e += 64) lds[(n1 + e / nrows4) * row_pad + e % nrows4] = 0.0;
    }
    wave_lds_sync();
    // Stage B: P = [J r]ᵀ [J r] on the matrix cores, upper 16×16 tiles (same scheme as the frame kernel: operand
    // element (col = 16t + (lane & 15), row = r0 + (lane >> 4)) serves as A of tile row t and as B of tile column t).
    if (it.rows_off >= 0) {
      // the rows go to the cell kernel, which forms [J r]ᵀ[J r] for all work items of the cell at once with four
      // waves: on the long single-lane chain of an IMU block this wave would spend another quarter of its time here
      double* dst = a.partials + it.rows_off;                  // 16-byte aligned (host)
      const int nw2 = (n1 * row_pad + 1) >> 1;                 // in double2 words; the staging area is at least that long
      const double2* s2 = reinterpret_cast<const double2*>(lds);
      double2* d2 = reinterpret_cast<double2*>(dst);
      for (int i = lane; i < nw2; i += 256) {                  // four 16-byte copies in flight per lane
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = s2[min(i + 64 * u, nw2 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + 64 * u < nw2) d2[i + 64 * u] = v[u];
      }
      ITICK(2)
      if (dbg) printf("eval_items cycles (kind %d, %d obs, %d cols, pad %d): setup+zero %lld  stage A %lld  rows out %lld\n", S.kind, it.obs_count, ncols, row_pad, tph[0], tph[1], tph[2]);
      return;
    }
    double* out = a.partials + it.partial_off;
    if (stage_only) { stage_only->nrows = nrows; stage_only->n1 = n1; stage_only->out = out; return; }     // (the workgroup shares the tiles out: eval_cells_kernel)
    stage_b_dispatch<0>(lds, row_pad, nrows, n1, out);
    ITICK(2)
    if (dbg) printf("eval_items cycles (kind %d, %d obs, %d cols, pad %d): setup+zero %lld  stage A %lld  stage B %lld\n", S.kind, it.obs_count, ncols, row_pad, tph[0], tph[1], tph[2]);
  }
#undef ITICK
}

// ---------------------------------------------------------------------------
// Camera frames (spline order 6). One wave per frame.
//
// All residual blocks of a frame share the pose p(t), its derivative and the
// spline weights w, and every Jacobian column of a block is a frame-constant linear
// combination of a few block-level columns (camera_cost_functor.h:71-147 by hand):
//   T = -(1/σ)√ρ' · D · R_rcᵀ   (2×3, D = d pixel / d x_c),   Y = T [y]×,   Z = T R_rw [R_wm x_m]×
//   pose: rotation Y·J_l(φ), position -T·R_rw;  camera extrinsics: q 2Y - 2T[t_rc]×, t -T;
//   body: q -2Z, t T·R_rw        (y: the point in the rig frame; Z only when the body pose is free)
// The wave stages only the SMALL prim columns [Y | T | (Z) | intrinsics | r] in LDS -- 15 for an
// 8-parameter camera with free extrinsics instead of 21: one 16-wide MFMA tile instead of three --
// and forms M_s = [..]ᵀ[..] over all blocks of the frame on the matrix cores. Once per frame,
// M = Bᵀ M_s B (B has at most three entries per column) is the block over the prim columns
//   [pose 6 | intrinsics | q | t | body q | body t | r],
// the latency row/column is -pdotᵀM, and M_ext with the expansion coefficients (spline weight of a
// column's control point) is the frame's compact record; T_frameᵀ M T_frame is expanded once per CELL.
// Every per-frame quantity (spline evaluation, Rodrigues terms, rotation products) is computed once
// per wave, not per block.
// ---------------------------------------------------------------------------
constexpr int kMaxPrim = 32;   // 6 + 11 + 3 + 3 + 3 + 3 + r = 30, padded to two 16-wide MFMA tiles
constexpr int kFramePad = 132; // row stride of a staged prim column: ≡ 4 (mod 32) spreads the MFMA operand read over all banks
constexpr int kMaxLocalCols = 96;

// Prim columns of a camera layout: [pose 6 | intrinsics | q | t | body q | body t | residual]; the latency column is
// the extra row/column PT of M_ext.
struct PrimMap { int intr, q, t, bq, bt, r, P1, PT, PE; };
DEV PrimMap prim_map(const LayoutDev& L, const SensorDev& S) {
  PrimMap m;
  int pc = 6;
  m.intr = L.c_intr >= 0 ? pc : -1; if (L.c_intr >= 0) pc += S.K;
  m.q = L.c_q >= 0 ? pc : -1; if (L.c_q >= 0) pc += 3;
  m.t = L.c_t >= 0 ? pc : -1; if (L.c_t >= 0) pc += 3;
  m.bq = L.c_bq >= 0 ? pc : -1; if (L.c_bq >= 0) pc += 3;
  m.bt = L.c_bt >= 0 ? pc : -1; if (L.c_bt >= 0) pc += 3;
  m.r = pc;
  m.P1 = pc + 1;                 // prim columns incl. residual
  m.PT = m.P1;                   // index of the latency row / column of M_ext (right behind the prim columns)
  m.PE = m.PT + 1;               // M_ext = [[M, Qᵀ], [Q, qq]]
  return m;
}
// Small prim columns (what is staged): [Y 3 | T 3 | Z 3 (body rotation free) | intrinsics | residual]
struct SmallMap { int z, k, r, P, PT; };
DEV SmallMap small_map(const LayoutDev& L, const SensorDev& S) {
  SmallMap m;
  int pc = 6;
  m.z = L.c_bq >= 0 ? pc : -1; if (L.c_bq >= 0) pc += 3;
  m.k = L.c_intr >= 0 ? pc : -1; if (L.c_intr >= 0) pc += S.K;
  m.r = pc;
  m.P = pc + 1;
  m.PT = (m.P + 15) & ~15;
  return m;
}
// prim column of local column lc of the layout (spline columns: the pose component)
DEV int prim_of_col(const LayoutDev& L, const SensorDev& S, const PrimMap& pm, int lc) {
  if (lc < 36) return lc % 6;
  if (lc == L.c_lat) return pm.PT;
  if (lc == L.ncols) return pm.r;
  if (L.c_intr >= 0 && lc >= L.c_intr && lc < L.c_intr + S.K) return pm.intr + (lc - L.c_intr);
  if (L.c_q >= 0 && lc >= L.c_q && lc < L.c_q + 3) return pm.q + (lc - L.c_q);
  if (L.c_t >= 0 && lc >= L.c_t && lc < L.c_t + 3) return pm.t + (lc - L.c_t);
  if (L.c_bq >= 0 && lc >= L.c_bq && lc < L.c_bq + 3) return pm.bq + (lc - L.c_bq);
  return pm.bt + (lc - L.c_bt);
}

template <int MODEL>
DEV bool frame_camera_block(const SensorDev& S, const double* intr, const M3& R_rc, const M3& R_rw, const M3& R_wm, V3 t_rc, V3 t_wm,
                            V3 t_wr, double px, double py, const double* xm, int apply_loss, double* Jp, int row0, int pc_z,
                            int pc_k, int pc_r, double* cost) {
  const V3 Rx = mul(R_wm, mk(xm[0], xm[1], xm[2]));
  const V3 y = mul(R_rw, (Rx + t_wm) - t_wr);
  const V3 z = y - t_rc;
  const V3 xc = mulT(R_rc, z);
  double pix[2], D[2][3], dK[2][kMaxIntr];
  if (!project<MODEL, true>(intr, xc, pix, D, dK)) return false;
  const double r0 = (px - pix[0]) * S.info, r1 = (py - pix[1]) * S.info;
  const double sq = r0 * r0 + r1 * r1;
  double ls = 1.0, rho = sq;
  if (apply_loss) rho = loss_eval(S.loss, S.loss_scale, sq, &ls);
  *cost = 0.5 * rho;
  const double fac = -S.info * ls;
  auto put = [&](int col, int r, double v) { Jp[col * kFramePad + row0 + r] = v; };
  put(pc_r, 0, r0 * ls); put(pc_r, 1, r1 * ls);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double T[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) T[j] = fac * (D[r][0] * R_rc.m[j][0] + D[r][1] * R_rc.m[j][1] + D[r][2] * R_rc.m[j][2]);
    put(0, r, T[1] * y.z - T[2] * y.y); put(1, r, T[2] * y.x - T[0] * y.z); put(2, r, T[0] * y.y - T[1] * y.x);   // Y = T [y]×
    put(3, r, T[0]); put(4, r, T[1]); put(5, r, T[2]);
    if (pc_z >= 0) {   // Z = (T R_rw) [R_wm x_m]×
      double G[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) G[j] = T[0] * R_rw.m[0][j] + T[1] * R_rw.m[1][j] + T[2] * R_rw.m[2][j];
      put(pc_z, r, G[1] * Rx.z - G[2] * Rx.y); put(pc_z + 1, r, G[2] * Rx.x - G[0] * Rx.z); put(pc_z + 2, r, G[0] * Rx.y - G[1] * Rx.x);
    }
  }
  if (pc_k >= 0) {
    constexpr int K = CamK<MODEL>::K;
#pragma unroll
    for (int j = 0; j < K; ++j) { put(pc_k + j, 0, fac * dK[0][j]); put(pc_k + j, 1, fac * dK[1][j]); }
  }
  return true;
}


// LDS of a frame workgroup (doubles): area A = the staged rows [Ps][kFramePad], later M_s, N and the column lists of B;
// then M_ext [PE][PE]; then the expansion coefficients [n1].
DEV int frame_area_a(int Ps, int PTs, int P1e) {
  const int after = PTs * PTs + ((Ps * P1e + 1) & ~1) + ((3 * P1e + 1) & ~1) + ((3 * P1e + 1) / 2 + 1);
  return (max((Ps + 1) * kFramePad, after) + 1) & ~1;      // (+ 1: the zero column, see eval_frames_body)
}

// value of `v` in lane `l` (l a compile-time constant after unrolling), wave-uniform
DEV double lane_value(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// What a frame's wave hands to its cell's workgroup (eval_cells_kernel) instead of writing a compact record: where M_ext and
// the expansion coefficients lie in its LDS area, and its share of the cell's pair table (requested behind the blocks' loop).
constexpr int kPairQ = 16;          // pairs per thread of a 128-thread workgroup: 2048 >= 63 * 64 / 2 (frame layouts: <= 60 columns)
struct FramePair {
  const double* Me; const double* coef;
  int PE, n1;
  int t0, stride;       // this thread's pairs: t0 + stride * q (128 threads of a two-frame cell's workgroup, or the 64 lanes of a solo frame)
  int e[kPairQ];
};
DEV void eval_frames_body(const EvalArgs& a, const int fidx, double* lds, FramePair* fp = nullptr) {
  if (a.st && (a.st->terminated || (a.need_flag && !a.st->need_jacobian))) return;
  const int lane = threadIdx.x & 63;
  const bool dbg = CAL_DEV_TIMING(a.debug && fidx == 7 && lane == 0);
  long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = dbg ? __builtin_readcyclecounter() : 0;
#define FTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  const FrameItemDev* ip = a.fitems + fidx;
  const FrameItemDev it = *ip;
  const LayoutDev& L = ip->L;       // (copies inside the frame record: one hop instead of frame -> layout -> sensor)
  const SensorDev& S = ip->S;
  constexpr int K = 6;
  const PrimMap pm = prim_map(L, S);
  const SmallMap sm = small_map(L, S);
  const int Kin = S.K;
  const int P1e = pm.P1, PT = pm.PT, PE = pm.PE;
  const int Ps = sm.P, PTs = sm.PT;
  const int ncols = L.ncols, n1 = ncols + 1;
  const int SA = frame_area_a(Ps, PTs, P1e);
  double* Jp = lds;                                  // [Ps][kFramePad]   staged rows, column-major
  double* Ms = lds;                                  // [PTs][PTs]        (after the last MFMA)
  double* Nmat = Ms + PTs * PTs;                     // [Ps][P1e]         M_s B
  double* bco = Nmat + ((Ps * P1e + 1) & ~1);        // [P1e][3]          column lists of B: coefficients ...
  int* bsrc = reinterpret_cast<int*>(bco + ((3 * P1e + 1) & ~1));   // [P1e][3]   ... and small prim columns
  double* Me = lds + SA;                             // [PE][PE]
  double* coef = Me + ((PE * PE + 1) & ~1);          // [n1] column c of the item = coef[c] · prim column prim[c]
  // observation of this lane in the first batch (clamped: loads stay unconditional), requested FIRST: the model point is a
  // dependent load (offset -> point), two round trips that run beside the frame's constants (round 5: requested behind
  // them, the first batch waited 2.9k clocks for its observations)
  auto obs_index = [&](int b0) { return it.obs_begin + min(b0 + lane, it.obs_count - 1); };
  int o_nx = obs_index(0);
  const double* xm_p = a.x + a.point_off[o_nx];
  double px_nx = a.m0[o_nx], py_nx = a.m1[o_nx];
  uint8_t act_nx = a.active ? a.active[o_nx] : uint8_t(1);     // outlier tag (nullptr: nothing is tagged), fetched ahead like the rest
  double xm_nx[3] = {xm_p[0], xm_p[1], xm_p[2]};
  // ---- per-frame quantities (every lane computes the same values) ----
  const int ki = it.seg + K - 1;
  const double* Mb = a.basis + size_t(it.seg) * K * K;
  const double lat = a.x[S.lat_off];
  double W[2][kMaxOrder];
  spline_weights<2, K>(K, a.knots[ki], a.knots[ki + 1], Mb, it.stamp - lat, W);
  double p[6] = {0, 0, 0, 0, 0, 0}, pd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const double* cp = a.x + ip->ctrl_off[i];
#pragma unroll
    for (int c = 0; c < 6; ++c) { p[c] += W[0][i] * cp[c]; pd[c] += W[1][i] * cp[c]; }
  }
  const double* qp = a.x + S.q_off; const double* tp = a.x + S.t_off;
  const double* bq = a.x + L.bq_off; const double* bt = a.x + L.bt_off;
  Q4 q_rc; q_rc.x = qp[0]; q_rc.y = qp[1]; q_rc.z = qp[2]; q_rc.w = qp[3];
  Q4 q_wm; q_wm.x = bq[0]; q_wm.y = bq[1]; q_wm.z = bq[2]; q_wm.w = bq[3];
  const M3 R_rc = rotmat(normalized(q_rc)), R_wm = rotmat(normalized(q_wm));
  const V3 t_rc = mk(tp[0], tp[1], tp[2]), t_wm = mk(bt[0], bt[1], bt[2]), t_wr = mk(p[3], p[4], p[5]);
  const V3 phi = mk(-p[0], -p[1], -p[2]);
  const M3 R_rw = rotmat(angle_axis_to_quat(phi));
  const M3 Jl = rod_J_matrix(rodrigues<double>(phi.x, phi.y, phi.z, false));
  const double* intr = a.x + S.intr_off;
  // expansion coefficients: item column lc = coef[lc] · prim column (spline columns carry their weight w_i)
  for (int lc = lane; lc < n1; lc += 64) {
    double cf = 1.0;
    if (lc < 36) {
      const int wi = lc / 6;
#pragma unroll
      for (int q = 0; q < K; ++q) cf = (wi == q) ? W[0][q] : cf;
    }
    coef[lc] = cf;
  }
  // ---- blocks -> staged rows -> M_s = [small prim columns]ᵀ[..] on the matrix cores ----
  // v_mfma_f64_16x16x4_f64: A[i][k] and B[k][j] of the product JᵀJ are the SAME staged value
  // Jp[col = 16t + (lane & 15)][row = r0 + (lane >> 4)], so one LDS read feeds both operands.
  const int lc16 = lane & 15, lk = lane >> 4;
  const bool two = PTs > 16;
  f64x4 acc00 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  // Round 5: NO arithmetic between the products. On this part a wave's v_mfma_f64 and its VALU instructions do not overlap
  // at all -- not only the FP64 ones: two v_cndmask in front of every product make it 103 clocks instead of 64
  // (profiles/microbench/mfma_f64_rate.hip) --, and the operands used to be masked one by one (columns past the layout's, rows
  // past the batch's): 72 products of a 144-block frame took 9.5-10k clocks. Now the lanes of columns that do not exist read
  // a column of zeros (staged column Ps, cleared once per frame) and the rows a short batch does not fill are cleared by the
  // lanes that have no block, so a group is eight loads and eight products.
  double* const zcol = Jp + Ps * kFramePad;
  for (int i = lane; i < kFramePad; i += 64) zcol[i] = 0.0;
  const double* op0 = (lc16 < Ps ? Jp + lc16 * kFramePad : zcol) + lk;
  const double* op1 = (16 + lc16 < Ps ? Jp + (16 + lc16) * kFramePad : zcol) + lk;
  double cost = 0.0, n_bad = 0.0;
  FTICK(0)
  for (int b0 = 0; b0 < it.obs_count; b0 += 64) {
    const int nb = min(64, it.obs_count - b0);
    wave_lds_sync();
    FTICK(1)
    const double px = px_nx, py = py_nx;
    const double xm[3] = {xm_nx[0], xm_nx[1], xm_nx[2]};
    const bool on = lane < nb && act_nx != 0;   // not tagged as an outlier
    if (b0 + 64 < it.obs_count) {   // wave-uniform: next batch's observation
      o_nx = obs_index(b0 + 64);
      px_nx = a.m0[o_nx]; py_nx = a.m1[o_nx];
      xm_p = a.x + a.point_off[o_nx];
      xm_nx[0] = xm_p[0]; xm_nx[1] = xm_p[1]; xm_nx[2] = xm_p[2];
      if (a.active) act_nx = a.active[o_nx];
    }
    // a tagged block, or (short batch) a lane without a block whose two rows lie in a group of 32 rows the products read: zeros
    // (a batch of 16 blocks -- the third of a 144-block frame -- fills its one group: nothing to clear)
    if (!on && lane < ((nb + 15) & ~15)) {
      for (int c = 0; c < Ps; ++c) { Jp[c * kFramePad + 2 * lane] = 0.0; Jp[c * kFramePad + 2 * lane + 1] = 0.0; }
    }
    if (on) {
      double c1 = 0.0;
      bool ok;
      switch (S.model) {
        case 1: ok = frame_camera_block<1>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
        case 2: ok = frame_camera_block<2>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
        case 3: ok = frame_camera_block<3>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
        case 4: ok = frame_camera_block<4>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
        case 5: ok = frame_camera_block<5>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
        case 6: ok = frame_camera_block<6>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
        default: ok = frame_camera_block<7>(S, intr, R_rc, R_rw, R_wm, t_rc, t_wm, t_wr, px, py, xm, a.apply_loss, Jp, 2 * lane, sm.z, sm.k, sm.r, &c1); break;
      }
      if (ok) cost += c1;
      else {   // an invalid block contributes nothing: zero its two rows
        n_bad += 1.0;
        for (int c = 0; c < Ps; ++c) { Jp[c * kFramePad + 2 * lane] = 0.0; Jp[c * kFramePad + 2 * lane + 1] = 0.0; }
      }
    }
    wave_lds_sync();
    FTICK(2)
    const int nrows = 2 * nb;
    // operands are fetched eight steps (32 rows) at a time so the LDS latency is paid once per group
    if (!two) {
      // one tile (<= 16 small prim columns: the usual case)
      // (requesting the next group's operands in front of this group's products was slower: the register moves it takes are
      //  VALU instructions again -- 8.0k clocks against 7.0k)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (32 * g < nrows) {
          double v0[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v0[u] = op0[32 * g + 4 * u];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0[u], v0[u], acc00, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (32 * g < nrows) {
        if (false) {
        } else {
          double v0[8], v1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { v0[u] = op0[32 * g + 4 * u]; v1[u] = op1[32 * g + 4 * u]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0[u], v0[u], acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[u], v0[u], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[u], v1[u], acc11, 0, 0, 0);
          }
        }
      }
    }
    }
    FTICK(3)
  }
  const double item_cost = wave_sum(cost);
  const double n_invalid = wave_sum(n_bad);
  if (lane == 0) { a.item_cost[2 * fidx] = item_cost; a.item_cost[2 * fidx + 1] = n_invalid; }
  if (fp) {      // this thread's pairs of the cell's expansion: on their way while the per-frame steps run
    const int n_pairs = n1 * (n1 + 1) / 2;
    const int* __restrict__ ptab = a.prim_tab + it.cell_prim_off;
#pragma unroll
    for (int q = 0; q < kPairQ; ++q) fp->e[q] = ptab[min(fp->t0 + fp->stride * q, n_pairs - 1)];
  }
  FTICK(6)
  // ---- M_s to LDS (full symmetric; C/D layout: col = lane & 15, row = (lane >> 4) + 4·reg) ----
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lk + 4 * r;
    Ms[row * PTs + lc16] = acc00[r];
    if (two) {
      Ms[(16 + row) * PTs + lc16] = acc10[r];
      Ms[lc16 * PTs + 16 + row] = acc10[r];
      Ms[(16 + row) * PTs + 16 + lc16] = acc11[r];
    }
  }
  // ---- M = Bᵀ M_s B over the prim columns [pose 6 | intrinsics | q | t | body q | body t | r] ----
  // Lane (h, j) = (lane >> 5, lane & 31) owns column j of B -- M-column = Σ_u co[u] · (small prim column src[u]), at most
  // three terms, kept in registers -- and works on every other row: N = M_s B column by column, then M = Bᵀ N through
  // M(j, i) = Σ_u co_j[u] · N(src_j[u], i), which needs nothing but the lane's own column again (M is symmetric). No
  // index division, no coefficient table in LDS, four rows in flight per lane.
  const int j = lane & 31, hh = lane >> 5;
  const bool jok = j < P1e;          // (P1e <= 31)
  int s0 = 0, s1 = 0, s2 = 0;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  // (column `jc` of a 3x3 matrix by selects: a run-time index into the matrix put both matrices into scratch -- the kernel's only
  //  scratch use, 152 bytes per lane and nine dependent scratch loads per frame)
  auto mcol = [](const M3& A, int r, int jc) { return jc == 0 ? A.m[r][0] : (jc == 1 ? A.m[r][1] : A.m[r][2]); };
  if (j < 3) { s0 = 0; s1 = 1; s2 = 2; c0 = mcol(Jl, 0, j); c1 = mcol(Jl, 1, j); c2 = mcol(Jl, 2, j); }          // rotation: Y·J_l
  else if (j < 6) { s0 = 3; s1 = 4; s2 = 5; c0 = -mcol(R_rw, 0, j - 3); c1 = -mcol(R_rw, 1, j - 3); c2 = -mcol(R_rw, 2, j - 3); }   // position: -T·R_rw
  else if (pm.intr >= 0 && j >= pm.intr && j < pm.intr + Kin) { s0 = sm.k + (j - pm.intr); c0 = 1.0; }
  else if (pm.q >= 0 && j >= pm.q && j < pm.q + 3) {                    // camera q: 2Y - 2T[t_rc]×
    const int c = j - pm.q, ka = (c + 1) % 3, kb = (c + 2) % 3;
    s0 = c; c0 = 2.0; s1 = 3 + ka; c1 = -2.0 * comp(t_rc, kb); s2 = 3 + kb; c2 = 2.0 * comp(t_rc, ka);
  }
  else if (pm.t >= 0 && j >= pm.t && j < pm.t + 3) { s0 = 3 + (j - pm.t); c0 = -1.0; }                 // camera t: -T
  else if (pm.bq >= 0 && j >= pm.bq && j < pm.bq + 3) { s0 = sm.z + (j - pm.bq); c0 = -2.0; }         // body q: -2Z
  else if (pm.bt >= 0 && j >= pm.bt && j < pm.bt + 3) { const int c = j - pm.bt; s0 = 3; s1 = 4; s2 = 5; c0 = mcol(R_rw, 0, c); c1 = mcol(R_rw, 1, c); c2 = mcol(R_rw, 2, c); }
  else { s0 = sm.r; c0 = 1.0; }
  FTICK(7)
  wave_lds_sync();
  if (jok) {                                                           // N = M_s B
    for (int ar = hh; ar < Ps; ar += 8) {
      double m0[4], m1[4], m2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a2 = min(ar + 2 * u, Ps - 1);
        m0[u] = Ms[a2 * PTs + s0]; m1[u] = Ms[a2 * PTs + s1]; m2[u] = Ms[a2 * PTs + s2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double sN = 0.0;
        sN += c0 * m0[u]; sN += c1 * m1[u]; sN += c2 * m2[u];
        if (ar + 2 * u < Ps) Nmat[(ar + 2 * u) * P1e + j] = sN;
      }
    }
  }
  FTICK(8)
  wave_lds_sync();
  if (jok) {                                                           // M = Bᵀ N
    for (int i = hh; i < P1e; i += 8) {
      double n0[4], n1v[4], n2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i2 = min(i + 2 * u, P1e - 1);
        n0[u] = Nmat[s0 * P1e + i2]; n1v[u] = Nmat[s1 * P1e + i2]; n2[u] = Nmat[s2 * P1e + i2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double sM = 0.0;
        sM += c0 * n0[u]; sM += c1 * n1v[u]; sM += c2 * n2[u];
        if (i + 2 * u < P1e) Me[j * PE + i + 2 * u] = sM;
      }
    }
  }
  FTICK(9)
  wave_lds_sync();
  {                                                                    // latency: dp/dlat = -pdot
    double sQ = 0.0;
    const int lq = min(lane, P1e - 1);
#pragma unroll
    for (int c = 0; c < 6; ++c) sQ -= pd[c] * Me[c * PE + lq];
    if (lane < P1e) { Me[PT * PE + lane] = sQ; Me[lane * PE + PT] = sQ; }
    double qq = 0.0;                                                   // (lanes 0..5 hold the entries the corner needs)
#pragma unroll
    for (int c = 0; c < 6; ++c) qq -= pd[c] * lane_value(sQ, c);
    if (lane == 0) Me[PT * PE + PT] = qq;
  }
  wave_lds_sync();
  FTICK(4)
  // ---- compact record: M_ext (PE×PE) then coef (n1). The expansion TᵀMT -- out(i, j) = coef_i coef_j M_ext(prim_i, prim_j)
  //      -- is done once per CELL by expand_cells_kernel over all its frames. Only prim rows / columns < P1e and the
  //      latency row / column PT are read there; the others are written as they lie. ----
  if (fp) {      // (the cell's workgroup expands out of LDS: no record)
    if (dbg) printf("eval_frames cycles (frame of %d blocks, %d small / %d prim cols, cell workgroup): frame-constants %lld  barrier %lld  blocks %lld  M-mfma %lld  M-to-lds [sums %lld  Ms+B %lld  N %lld  M %lld  latency %lld]\n",
                    it.obs_count, Ps, P1e, tph[0], tph[1], tph[2], tph[3], tph[6], tph[7], tph[8], tph[9], tph[4]);
    fp->Me = Me; fp->coef = coef; fp->PE = PE; fp->n1 = n1; return;
  }
  double* out = a.partials + it.partial_off;
  const int nme = PE * PE;
  for (int i = lane; i < nme; i += 64) out[i] = Me[i];
  for (int i = lane; i < n1; i += 64) out[nme + i] = coef[i];
  FTICK(5)
  if (dbg) printf("eval_frames cycles (frame of %d blocks, %d small / %d prim cols): frame-constants %lld  barrier %lld  blocks %lld  M-mfma %lld  M-to-lds [sums %lld  Ms+B %lld  N %lld  M %lld  latency %lld]  record %lld\n",
                  it.obs_count, Ps, P1e, tph[0], tph[1], tph[2], tph[3], tph[6], tph[7], tph[8], tph[9], tph[4], tph[5]);
#undef FTICK
}

template <bool JAC, int KT>
__global__ __launch_bounds__(64) void eval_items_kernel(EvalArgs a) {
  extern __shared__ double lds[];
  eval_items_body<JAC, KT>(a, blockIdx.x, lds);
}

__global__ __launch_bounds__(64) void eval_frames_kernel(EvalArgs a) {
  extern __shared__ double lds[];
  eval_frames_body(a, blockIdx.x, lds);
}

// Whole Jacobian pass in one launch (spline order 6): the generic items (IMU cells; they are few and each is a long
// single-wave computation, so they go first) and the camera frames run side by side instead of back to back.
// The extra workgroup of the Jacobian launch in the streaming solve loop (EvalArgs.hint_progress): will the control stage
// behind this evaluation end the solve? By the parameter tolerance: the very test of control_body on the very sums. By the
// function tolerance: the model's cost change stands in for the candidate's (they agree to a few per cent where a solve
// converges). "Go" -> the host enqueues the next iteration now, while this evaluation runs; otherwise it waits for the
// control stage's word, and a solve that does end leaves nothing behind it on the stream. A wrong "hold" costs the host's
// reaction time once, a wrong "go" what every solve cost before; the results do not depend on either.
DEV void end_hint_body(const EvalArgs& a) {
  const LmState* st = a.st;
  const int lane = threadIdx.x;
  if (st->terminated) return;
  const int parts = st->upd_parts;
  const double* upd = st->upd_ext;
  const int n = st->upd_ext_n;
  const double x_cost = st->x_cost, x_norm = st->x_norm;
  const int chol_failed = st->chol_failed, epoch = st->sink.epoch;
  double mcc = 0.0, sn = 0.0, bad = 0.0;
  if (parts == 0 && upd) {
    for (int i = lane; i < n; i += 64) { mcc += upd[4 * i]; sn += upd[4 * i + 1]; bad += upd[4 * i + 3]; }
  } else if (lane < parts) {
    mcc = st->upd_mcc[lane]; sn = st->upd_sn[lane]; bad = st->upd_bad[lane] ? 1.0 : 0.0;
  }
  mcc = wave_sum(mcc); sn = wave_sum(sn); bad = wave_sum(bad);
  bool end = false;
  if (!(bad > 0.0) && !chol_failed && mcc > 0.0)
    end = sqrt(sn) <= a.hint_ptol * (x_norm + a.hint_ptol) || mcc <= 1.25 * a.hint_ftol * x_cost;
  if (!end && lane == 0)
    __hip_atomic_store(a.hint_progress + 2, (epoch << 20) | a.hint_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void eval_jacobian_kernel(EvalArgs a) {
  extern __shared__ double lds[];
  if (int(blockIdx.x) == a.n_items + a.n_fitems) { end_hint_body(a); return; }
  const unsigned long long t0 = CAL_DEV_TIMING(a.debug >= 3) ? __builtin_amdgcn_s_memrealtime() : 0;
  if (int(blockIdx.x) < a.n_items) eval_items_body<true, 6>(a, blockIdx.x, lds);
  else eval_frames_body(a, blockIdx.x - a.n_items, lds);
  if (CAL_DEV_TIMING(a.debug >= 3) && a.wave_log && threadIdx.x == 0) {   // CALICO_KERNEL_TIMING=3: life span of every wave (100 MHz clock)
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (t1 - t0 > 200) { a.wave_log[2 * blockIdx.x] = t0; a.wave_log[2 * blockIdx.x + 1] = t1; }   // not the early exits after termination
  }
}

// Cell workgroups (EvalArgs.pair_mode): the Jacobian pass in workgroups of TWO waves, one per SIMD pair of a CU.
//   workgroups [0, n_item_wg): two work items (IMU cells), each wave its own -- they form their blocks themselves;
//   workgroups behind them: wave w evaluates frame 2 g + w (a.fitems holds two entries per workgroup). The two frames of ONE
//     cell: M_ext and the expansion coefficients stay in the waves' LDS areas, and behind ONE workgroup barrier both waves expand
//     the cell's block, pair by pair, frame 0 then frame 1 -- the sums of expand_cells_kernel in the same order (bit-identical)
//     without the compact record's trip through memory, without that kernel's launch and without its dependent loads (cell
//     descriptor -> records). Or two one-frame cells ("solo"): each wave expands its own block out of its own LDS area;
//   one more workgroup for the end hint.
// 2 x (cells + item pairs) waves: what the one-wave launch had, in the same single round of one wave per SIMD.
DEV void eval_cells_body(const EvalArgs& a, double* lds);
__global__ __launch_bounds__(128) void eval_cells_kernel(EvalArgs a) {
  extern __shared__ double lds[];
  const unsigned long long t0 = CAL_DEV_TIMING(a.debug >= 3) ? __builtin_amdgcn_s_memrealtime() : 0;
  eval_cells_body(a, lds);
  if (CAL_DEV_TIMING(a.debug >= 3) && a.wave_log && (threadIdx.x & 63) == 0) {   // CALICO_KERNEL_TIMING=3: life span of every wave (100 MHz clock)
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const int w = 2 * blockIdx.x + (threadIdx.x >> 6);
    if (t1 - t0 > 200) { a.wave_log[2 * w] = t0; a.wave_log[2 * w + 1] = t1; }
  }
}
DEV void eval_cells_body(const EvalArgs& a, double* lds) {
  const int tid = threadIdx.x, wave = tid >> 6;
  double* const lds_w = lds + size_t(wave) * a.wave_lds_doubles;
  const int n_item_wg = (a.n_items + 1) >> 1, n_cell_wg = a.n_fitems >> 1;
  const int hint_wg = a.hint_progress && a.st && a.hint_first ? 1 : 0;       // (launch_eval_jacobian: one more workgroup)
  if (hint_wg && blockIdx.x == 0) { if (wave == 0) end_hint_body(a); return; }
  const int g = int(blockIdx.x) - hint_wg;
  if (g < n_item_wg) {
    // items g and g + n_item_wg: a gyroscope cell and an accelerometer cell where the problem has both (the items come sorted
    // by sensor). The accelerometer's rows are staged ~4 us after the gyroscope's and [J r]ᵀ[J r] is 10 tiles x 16 MFMAs for
    // either: the early wave forms all of its own tiles, then the last four of the late item's ten, while the late wave forms the
    // first six (a tile's k-steps stay in one wave: the sums do not change).
    if (a.st && (a.st->terminated || (a.need_flag && !a.st->need_jacobian))) return;
    const int i0 = g, i1 = g + n_item_wg;
    const bool both = i1 < a.n_items;
    const int mine = wave == 0 ? i0 : i1, other = wave == 0 ? i1 : i0;
    if (!both) { if (wave == 0) eval_items_body<true, 6>(a, i0, lds_w); return; }
    const bool late_m = a.items[mine].S.kind == 2, late_o = a.items[other].S.kind == 2;      // (CALICO_SENSOR_ACCELEROMETER)
    const bool shared = late_m != late_o && a.items[mine].rows_off < 0 && a.items[other].rows_off < 0;
    if (!shared) { eval_items_body<true, 6>(a, mine, lds_w); return; }
    const bool dbg = CAL_DEV_TIMING(a.debug == 1 && g == 3 && (tid & 63) == 0);
    const long long tq0 = dbg ? __builtin_readcyclecounter() : 0;
    long long tq1 = 0, tq2 = 0, tq3 = 0;
    // (the late wave does not wait for the early one: it raises a word in LDS when its rows are staged -- the word is cleared
    //  in front of a barrier both waves pass at once -- and goes on with its share; the early wave looks at the word when its own
    //  block is done)
    int* const staged = reinterpret_cast<int*>(lds + 2 * size_t(a.wave_lds_doubles));
    if (tid == 0) __hip_atomic_store(staged, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    ItemStage st;
    eval_items_body<true, 6>(a, mine, lds_w, &st);
    if (dbg) tq1 = __builtin_readcyclecounter();
    if (!late_m) {
      stage_b_dispatch<0>(lds_w, a.row_pad, st.nrows, st.n1, st.out);
      if (dbg) tq2 = __builtin_readcyclecounter();
      // (LDS executes a wave's accesses in order and the late wave drains its stores in front of the flag's: once the flag
      //  is seen the rows are there. The compiler must not move the rows' reads in front of the loop: atomic load + barrier
      //  for the compiler, like elim_follow's channel reads in block_elim.hpp)
      while (__hip_atomic_load(staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      if (dbg) tq3 = __builtin_readcyclecounter();
      const ItemDev* op = a.items + other;
      const int n1o = op->L.ncols + 1, nro = (op->S.kind == 0 ? 2 : 3) * op->obs_count;
      stage_b_dispatch<2>(lds + size_t(wave ^ 1) * a.wave_lds_doubles, a.row_pad, nro, n1o, a.partials + op->partial_off);
    } else {
      if (dbg) tq2 = __builtin_readcyclecounter();
      wave_lds_sync();                       // (the rows are in LDS)
      asm volatile("" ::: "memory");
      if ((tid & 63) == 0) __hip_atomic_store(staged, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (dbg) tq3 = __builtin_readcyclecounter();
      stage_b_dispatch<1>(lds_w, a.row_pad, st.nrows, st.n1, st.out);
    }
    if (dbg) printf("item pair wave %d (late %d): rows staged after %lld clocks, own tiles %lld, barrier %lld, shared tiles %lld\n", wave, int(late_m), tq1 - tq0, tq2 - tq1, tq3 - tq2, (long long)__builtin_readcyclecounter() - tq3);
    return;
  }
  if (g >= n_item_wg + n_cell_wg) { if (wave == 0 && a.hint_progress) end_hint_body(a); return; }
  if (a.st && (a.st->terminated || (a.need_flag && !a.st->need_jacobian))) return;      // (both waves alike: nobody is left at the barrier)
  const int fidx = 2 * (g - n_item_wg) + wave;
  const FrameItemDev* ip = a.fitems + fidx;
  const bool solo = ip->cell_pad != 0;       // (both entries of a workgroup are of one kind: build_plan)
  FramePair fp;
  fp.Me = nullptr; fp.coef = nullptr; fp.PE = 0; fp.n1 = 0;
  if (solo && ip->obs_count == 0) { if ((tid & 63) == 0) { a.item_cost[2 * fidx] = 0.0; a.item_cost[2 * fidx + 1] = 0.0; } return; }
  fp.t0 = solo ? (tid & 63) : tid; fp.stride = solo ? 64 : 128;
  eval_frames_body(a, fidx, lds_w, &fp);       // (one call site: inlined twice, the kernel spilled twice as much)
  if (solo) {
    // a one-frame cell: the wave expands its own block, straight out of its LDS area (the same products as expand_cells_kernel
    // forms for a cell of one frame); the other wave of the workgroup does the same for another cell, or has nothing to do
    const int lane = tid & 63;
    wave_lds_sync();
    const int n1 = fp.n1, n_pairs = n1 * (n1 + 1) / 2;
    const int* __restrict__ ptab = a.prim_tab + ip->cell_prim_off;
    int e2[kPairQ];
#pragma unroll
    for (int q = 0; q < kPairQ; ++q) e2[q] = ptab[min(lane + 64 * (kPairQ + q), n_pairs - 1)];      // (the second half of the lane's pairs)
    double* out = a.partials + ip->cell_partial_off;
#pragma unroll
    for (int q = 0; q < kPairQ; ++q) {
      if (lane + 64 * q >= n_pairs) continue;
      const int pi = fp.e[q] & 255, pj = (fp.e[q] >> 8) & 255, pm_off = fp.e[q] >> 16;
      double acc = 0.0;
      acc += fp.coef[pi] * fp.coef[pj] * fp.Me[pm_off];
      block_store(out + tri_off(pi, pj, n1), acc);
    }
#pragma unroll
    for (int q = 0; q < kPairQ; ++q) {
      if (lane + 64 * (kPairQ + q) >= n_pairs) continue;
      const int pi = e2[q] & 255, pj = (e2[q] >> 8) & 255, pm_off = e2[q] >> 16;
      double acc = 0.0;
      acc += fp.coef[pi] * fp.coef[pj] * fp.Me[pm_off];
      block_store(out + tri_off(pi, pj, n1), acc);
    }
    return;
  }
  __syncthreads();
  const int n1 = fp.n1, n_pairs = n1 * (n1 + 1) / 2;
  // frame f's M_ext / coefficients: in wave f's area, at the offsets this wave's own have in its area (one layout per cell)
  const long long shift = (long long)a.wave_lds_doubles;
  const double* const me0 = fp.Me - size_t(wave) * shift;
  const double* const cf0 = fp.coef - size_t(wave) * shift;
  double* out = a.partials + ip->cell_partial_off;
#pragma unroll
  for (int q = 0; q < kPairQ; ++q) {
    if (tid + 128 * q >= n_pairs) continue;
    const int pi = fp.e[q] & 255, pj = (fp.e[q] >> 8) & 255, pm_off = fp.e[q] >> 16;
    double acc = 0.0;
    acc += cf0[pi] * cf0[pj] * me0[pm_off];
    acc += cf0[shift + pi] * cf0[shift + pj] * me0[shift + pm_off];
    block_store(out + tri_off(pi, pj, n1), acc);
  }
}

// Expansion + sum of the compact frame records of one cell: block (i, j), i <= j, of the cell's partial is
//   Σ_frames coef_f(i) · coef_f(j) · M_ext,f(prim_i, prim_j)   in frame order (deterministic).
// One workgroup per cell; records are staged through LDS in chunks of a.cell_chunk frames. Everything the kernel
// needs sits in the cell descriptor and a per-layout table: a chain of dependent global loads (item -> layout ->
// sensor -> ...) costs more than the arithmetic here.
// [J r]ᵀ[J r] of a cell whose work items filed their staged rows (IMU): the upper 16×16 tiles are dealt to the four
// waves, every wave walks the items of the cell in order (rows in order: deterministic) with its tiles' accumulators
// in registers; the row stores pass through LDS a.row_cell_chunk items at a time.
template <int MAXT>
DEV void row_cell_body(const EvalArgs& a, const CellDev& cell, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc16 = lane & 15, lk = lane >> 4;
  const int n1 = cell.n1, NT = (n1 + 15) >> 4, ntile = NT * (NT + 1) / 2;
  const int pad = a.row_pad, stride = (a.lds_cols * a.row_pad + 1) & ~1, words = n1 * pad;   // stride: see finalize (even)
  // tiles t = wave, wave + 4, ...: (I, J), I <= J, in row-major order of the upper triangle
  int tI[MAXT], tJ[MAXT];
  f64x4 acc[MAXT];
  {
    int t = 0, I = 0, J = 0;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
      const int want = wave + 4 * q;
      while (t < want && I < NT) { ++t; if (++J == NT) { ++I; J = I; } }
      tI[q] = want < ntile ? I : 0; tJ[q] = want < ntile ? J : 0;
      acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
  }
  const double* src = a.partials + cell.src_off;
  const bool dbg = CAL_DEV_TIMING(a.debug && int(blockIdx.x) == a.n_cells - 3 && lane == 0);
  long long tph[4] = {0, 0, 0, 0}, tk = dbg ? __builtin_readcyclecounter() : 0;
#define RTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  for (int i0 = 0; i0 < cell.frame_count; i0 += a.row_cell_chunk) {
    const int ni = min(a.row_cell_chunk, cell.frame_count - i0);
    __syncthreads();
    RTICK(0)
    for (int it = 0; it < ni; ++it) {
      const double* s = src + size_t(i0 + it) * stride;
      // sixteen loads in flight per thread: a load-store loop waits out the full memory latency on every trip
      for (int b = tid; b < words; b += 256 * 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s[min(b + 256 * u, words - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (b + 256 * u < words) lds[it * words + b + 256 * u] = v[u];
      }
    }
    __syncthreads();
    RTICK(1)
    for (int it = 0; it < ni; ++it) {
      const int nrows = min(cell.PE, cell.pad0 - (i0 + it) * cell.PE);
      const double* base = lds + it * words;
      // column offsets and masks of this wave's tiles do not depend on the row: hoisted; KU k-steps (4·KU rows) of
      // operands are requested before the first of their MFMAs
      int oci[MAXT], ocj[MAXT];
      double mi[MAXT], mj[MAXT];
#pragma unroll
      for (int q = 0; q < MAXT; ++q) {
        const int ci = 16 * tI[q] + lc16, cj = 16 * tJ[q] + lc16;
        oci[q] = (ci < n1 ? ci : n1 - 1) * pad; ocj[q] = (cj < n1 ? cj : n1 - 1) * pad;
        mi[q] = ci < n1 ? 1.0 : 0.0; mj[q] = cj < n1 ? 1.0 : 0.0;
      }
      constexpr int KU = MAXT <= 3 ? 4 : 1;      // (seven tiles per wave: the registers hold one k-step of operands)
      for (int r0 = 0; r0 < nrows; r0 += 4 * KU) {
        double oi[KU][MAXT], oj[KU][MAXT];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int r = r0 + 4 * u + lk;
          const int rc = r < nrows ? r : nrows - 1;
          const double rm = r < nrows ? 1.0 : 0.0;
#pragma unroll
          for (int q = 0; q < MAXT; ++q) { oi[u][q] = base[oci[q] + rc] * (mi[q] * rm); oj[u][q] = base[ocj[q] + rc] * mj[q]; }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
          for (int q = 0; q < MAXT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(oi[u][q], oj[u][q], acc[q], 0, 0, 0);
      }
    }
  }
  RTICK(2)
  if (dbg) printf("row_cell cycles (wave %d, %d items, %d rows, n1 %d): setup %lld  staging %lld  mfma %lld\n", wave, cell.frame_count, cell.pad0, n1, tph[0], tph[1], tph[2]);
#undef RTICK
  double* out = a.partials + cell.partial_off;
#pragma unroll
  for (int q = 0; q < MAXT; ++q) {
    if (wave + 4 * q >= ntile) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = 16 * tI[q] + lk + 4 * r, gj = 16 * tJ[q] + lc16;
      if (gi <= gj && gj < n1) out[tri_off(gi, gj, n1)] = acc[q][r];
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void expand_cells_kernel(EvalArgs a) {
  extern __shared__ double lds[];
  const long long t_start = __builtin_readcyclecounter();
  const CellDev cell = a.cells[blockIdx.x];
  if (a.st && (a.st->terminated || (a.need_flag && !a.st->need_jacobian))) return;
  if (cell.prim_off < 0) {
    if (cell.n1 <= 64) row_cell_body<3>(a, cell, lds); else row_cell_body<7>(a, cell, lds);
    return;
  }
  const int tid = threadIdx.x;
  const bool dbg = CAL_DEV_TIMING(a.debug && blockIdx.x == 5 && tid == 0);
  long long tph[5] = {0, 0, 0, 0, 0}, tk = t_start;
#define CTICK(i) if (dbg) { const long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; }
  const int n1 = cell.n1, PE = cell.PE, nme = PE * PE, rec = nme + n1;
  const int n_pairs = n1 * (n1 + 1) / 2;
  // per-layout pair table (host): entry t of the row-major upper triangle = i | j << 8 | (prim_i·PE + prim_j) << 16.
  // One coalesced load per pair, all issued together (decoding the pairs in the kernel took data-dependent loops and
  // two more dependent table loads).
  const int* __restrict__ ptab = a.prim_tab + cell.prim_off;
  constexpr int NQ = 12;             // pairs per thread: n1 <= 77
  int pi[NQ], pj[NQ], pm_off[NQ];
  double acc[NQ];
  // the first chunk of records is requested together with the pair table (both only need the cell descriptor)
  const double* cell_src0 = a.partials + cell.src_off;
  const int nf0 = min(a.cell_chunk, cell.frame_count), nw0 = nf0 * rec;
  double rv[12];
#pragma unroll
  for (int u = 0; u < 12; ++u) rv[u] = cell_src0[min(tid + 256 * u, nw0 - 1)];
  {
    int e[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) e[q] = ptab[min(tid + 256 * q, n_pairs - 1)];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { pi[q] = e[q] & 255; pj[q] = (e[q] >> 8) & 255; pm_off[q] = e[q] >> 16; acc[q] = 0.0; }
  }
  CTICK(0)
  const double* cell_src = a.partials + cell.src_off;
  for (int f0 = 0; f0 < cell.frame_count; f0 += a.cell_chunk) {
    const int nf = min(a.cell_chunk, cell.frame_count - f0);
    __syncthreads();
    const double* src = cell_src + size_t(f0) * rec;
    if (f0 == 0) {
#pragma unroll
      for (int u = 0; u < 12; ++u) if (tid + 256 * u < nw0) lds[tid + 256 * u] = rv[u];
      for (int i = tid + 256 * 12; i < nf * rec; i += 256) lds[i] = src[i];
    } else {
      for (int i = tid; i < nf * rec; i += 256) lds[i] = src[i];
    }
    __syncthreads();
    CTICK(1)
    for (int f = 0; f < nf; ++f) {
      const double* me = lds + f * rec;
      const double* cf = me + nme;
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] += cf[pi[q]] * cf[pj[q]] * me[pm_off[q]];
    }
  }
  CTICK(2)
  double* out = a.partials + cell.partial_off;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (tid + 256 * q < n_pairs) out[tri_off(pi[q], pj[q], n1)] = acc[q];
  CTICK(3)
  if (dbg) printf("expand_cells cycles (cell of %d frames, n1 %d): setup+tables %lld  record copy %lld  accumulate %lld  store %lld\n",
                  cell.frame_count, n1, tph[0], tph[1], tph[2], tph[3]);
#undef CTICK
}

void launch_expand_cells(const EvalArgs& a, hipStream_t stream) {
  if (a.n_cells == 0) return;
  const size_t frame_bytes = size_t(a.cell_chunk) * a.cell_rec_max * sizeof(double);
  const size_t row_bytes = size_t(a.row_cell_chunk) * a.lds_cols * a.row_pad * sizeof(double);
  hipLaunchKernelGGL(expand_cells_kernel, dim3(a.n_cells), dim3(256), std::max(frame_bytes, row_bytes), stream, a);
}

// Outlier tagging on the device (the notebooks' loop: residual norm > tau -> MarkOutliersById): observations
// [begin, end) of one sensor in sorted order; res/valid come from the residual write-back without loss.
__global__ void mark_outliers_kernel(const double* __restrict__ res, const uint8_t* __restrict__ valid, uint8_t* active,
                                     int begin, int end, int dim, double threshold, int* n_marked) {
  const int q = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= end || !active[q]) return;
  double sq = 0.0;
  for (int c = 0; c < dim; ++c) sq += res[size_t(q) * 3 + c] * res[size_t(q) * 3 + c];
  const bool inlier = valid[q] && sqrt(sq) <= threshold;
  if (!inlier) { active[q] = 0; atomicAdd(n_marked, 1); }
}
// Inlier test of the residual pairs on the device (sensor_base.h GetMeasurementResidualPairs + the notebooks'
// `norm <= tau` filter): the mask replaces the validity byte of every observation in [begin, end) (sorted order).
__global__ void inlier_mask_kernel(const double* __restrict__ res, uint8_t* valid_then_mask, const uint8_t* __restrict__ active,
                                   int begin, int end, int dim, double threshold) {
  const int q = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= end) return;
  double sq = 0.0;
  for (int c = 0; c < dim; ++c) sq += res[size_t(q) * 3 + c] * res[size_t(q) * 3 + c];
  valid_then_mask[q] = (active[q] && valid_then_mask[q] && sqrt(sq) <= threshold) ? 1 : 0;
}
void launch_inlier_mask(const double* res, uint8_t* valid_then_mask, const uint8_t* active, int begin, int end, int dim, double threshold,
                        hipStream_t s) {
  if (end > begin) hipLaunchKernelGGL(inlier_mask_kernel, dim3((end - begin + 255) / 256), dim3(256), 0, s, res, valid_then_mask, active, begin, end, dim, threshold);
}
// Residual statistics per image region: one workgroup per bin walks the sensor's observations (sorted order
// [begin, end)) in a fixed thread-strided order and reduces in a fixed tree -- deterministic, no atomics.
__global__ __launch_bounds__(256) void residual_heatmap_kernel(const double* __restrict__ res, const uint8_t* __restrict__ valid,
                                                               const uint8_t* __restrict__ active, const double* __restrict__ px,
                                                               const double* __restrict__ py, int begin, int end, double img_w,
                                                               double img_h, int num_rows, int num_cols, double* rmse,
                                                               long long* count) {
  __shared__ double s_sq[256];
  __shared__ long long s_n[256];
  const int bin = blockIdx.x, brow = bin / num_cols, bcol = bin % num_cols, tid = threadIdx.x;
  double sq = 0.0;
  long long n = 0;
  for (int o = begin + tid; o < end; o += 256) {
    if ((active && !active[o]) || !valid[o]) continue;
    int c = int(floor(px[o] / img_w * num_cols)), r = int(floor(py[o] / img_h * num_rows));   // operation order of utils.py:39-40
    c = max(min(c, num_cols - 1), 0); r = max(min(r, num_rows - 1), 0);
    if (r != brow || c != bcol) continue;
    const double r0 = res[size_t(o) * 3], r1 = res[size_t(o) * 3 + 1];
    sq += r0 * r0 + r1 * r1; n += 1;
  }
  s_sq[tid] = sq; s_n[tid] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_sq[tid] += s_sq[tid + off]; s_n[tid] += s_n[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { count[bin] = s_n[0]; rmse[bin] = sqrt(s_sq[0] / double(s_n[0])); }   // 0/0 = NaN for an empty bin
}
void launch_residual_heatmap(const double* res, const uint8_t* valid, const uint8_t* active, const double* px, const double* py,
                             int begin, int end, int width, int height, int num_rows, int num_cols, double* rmse, long long* count,
                             hipStream_t s) {
  hipLaunchKernelGGL(residual_heatmap_kernel, dim3(num_rows * num_cols), dim3(256), 0, s, res, valid, active, px, py, begin, end,
                     double(width), double(height), num_rows, num_cols, rmse, count);
}

void launch_mark_outliers(const double* res, const uint8_t* valid, uint8_t* active, int begin, int end, int dim,
                          double threshold, int* n_marked, hipStream_t s) {
  if (end > begin) hipLaunchKernelGGL(mark_outliers_kernel, dim3((end - begin + 255) / 256), dim3(256), 0, s, res, valid, active, begin, end, dim, threshold, n_marked);
}

// LDS doubles of a frame workgroup whose layout has Ps small prim columns, P1e prim columns (residual included both) and
// n1 local columns; the worst case over all layouts the frame kernel takes (P1e <= 31, Ps <= 25)
size_t frame_lds_doubles(int Ps, int P1e, int n1) {
  const int PTs = (Ps + 15) & ~15, PE = P1e + 1;
  const int after = PTs * PTs + ((Ps * P1e + 1) & ~1) + ((3 * P1e + 1) & ~1) + ((3 * P1e + 1) / 2 + 1);
  const int SA = (std::max((Ps + 1) * kFramePad, after) + 1) & ~1;
  return size_t(SA) + size_t((PE * PE + 1) & ~1) + size_t(n1) + 16;
}
size_t frame_lds_bytes() { return frame_lds_doubles(25, 31, kMaxLocalCols) * sizeof(double); }
static size_t frame_launch_bytes(const EvalArgs& a) {
  return a.frame_lds_doubles > 0 ? size_t(a.frame_lds_doubles) * sizeof(double) : frame_lds_bytes();
}
void launch_eval_frames(const EvalArgs& a, hipStream_t stream) {
  if (a.n_fitems == 0) return;
  hipLaunchKernelGGL(eval_frames_kernel, dim3(a.n_fitems), dim3(64), frame_launch_bytes(a), stream, a);
}

// items (a.items / a.n_items, cost slots from a.cost_index_base) and frames (a.fitems / a.n_fitems) together
void launch_eval_jacobian(const EvalArgs& a, hipStream_t stream) {
  if (a.n_items + a.n_fitems == 0) return;
  if (a.pair_mode) {
    const int hint = a.hint_progress && a.st ? 1 : 0;
    hipLaunchKernelGGL(eval_cells_kernel, dim3(((a.n_items + 1) >> 1) + (a.n_fitems >> 1) + hint), dim3(128),
                       cells_launch_lds_bytes(size_t(a.wave_lds_doubles)), stream, a);
    return;
  }
  size_t lds = size_t(a.lds_cols) * a.row_pad * sizeof(double);
  if (a.n_fitems > 0 && lds < frame_launch_bytes(a)) lds = frame_launch_bytes(a);
  const int hint = a.hint_progress && a.st ? 1 : 0;      // one more workgroup: end_hint_body
  hipLaunchKernelGGL(eval_jacobian_kernel, dim3(a.n_items + a.n_fitems + hint), dim3(64), lds, stream, a);
}

void launch_eval(const EvalArgs& a, bool jac, hipStream_t stream) {
  if (a.n_items == 0) return;
  if (jac) {
    const size_t lds = size_t(a.lds_cols) * a.row_pad * sizeof(double);
    if (a.order == 6) hipLaunchKernelGGL((eval_items_kernel<true, 6>), dim3(a.n_items), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((eval_items_kernel<true, 0>), dim3(a.n_items), dim3(64), lds, stream, a);
  } else {
    if (a.order == 6) hipLaunchKernelGGL((eval_items_kernel<false, 6>), dim3(a.n_items), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((eval_items_kernel<false, 0>), dim3(a.n_items), dim3(64), 0, stream, a);
  }
}

hipError_t configure_eval_kernels(size_t max_lds_bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_frames_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, int(frame_lds_bytes()));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&expand_cells_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_cells_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kCellsMaxLds));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_jacobian_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          int(std::max<size_t>(std::max(max_lds_bytes, frame_lds_bytes()), 80 * 1024)));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_items_kernel<true, 6>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(max_lds_bytes));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_items_kernel<true, 0>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(max_lds_bytes));
}

}  // namespace cal
