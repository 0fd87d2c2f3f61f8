// device_math.hpp — gfx950 device functions for the calibration residuals.
//
// FP64 throughout (the reference path is double only). Each function cites the
// reference arithmetic it reproduces (paths relative to the reference tree).
// Jacobians are hand-derived (chain rule through pose -> point -> pixel);
// only the Rodrigues-Jacobian pieces of the IMU residuals, whose closed-form
// derivative is third order, use a 3-lane forward dual (D3) in registers.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DEV __device__ __forceinline__
#else
// Plain host build (include/calico/calico.hpp uses the same model code for Sensor::Project).
#include <cmath>
#define DEV inline
namespace cal { using std::sqrt; using std::sin; using std::cos; using std::tan; using std::atan; using std::log; using std::fmax; }
#endif

namespace cal {

struct V3 { double x, y, z; };
struct M3 { double m[3][3]; };
struct Q4 { double x, y, z, w; };  // Eigen coeffs() order

DEV V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV double comp(const V3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

DEV M3 skew(V3 v) {
  M3 r;
  r.m[0][0] = 0; r.m[0][1] = -v.z; r.m[0][2] = v.y;
  r.m[1][0] = v.z; r.m[1][1] = 0; r.m[1][2] = -v.x;
  r.m[2][0] = -v.y; r.m[2][1] = v.x; r.m[2][2] = 0;
  return r;
}
DEV M3 mul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
DEV M3 transpose(const M3& a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
DEV V3 mul(const M3& a, V3 v) {
  return mk(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
// aᵀ v
DEV V3 mulT(const M3& a, V3 v) {
  return mk(a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z, a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
            a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z);
}

// Rotation matrix of a (near-unit) quaternion, consistent with Eigen's
// q * v = v + 2w(u×v) + 2u×(u×v)  (typedefs.h / Eigen _transformVector).
DEV M3 rotmat(Q4 q) {
  const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  M3 r;
  r.m[0][0] = 1.0 - 2.0 * (yy + zz); r.m[0][1] = 2.0 * (xy - wz); r.m[0][2] = 2.0 * (xz + wy);
  r.m[1][0] = 2.0 * (xy + wz); r.m[1][1] = 1.0 - 2.0 * (xx + zz); r.m[1][2] = 2.0 * (yz - wx);
  r.m[2][0] = 2.0 * (xz - wy); r.m[2][1] = 2.0 * (yz + wx); r.m[2][2] = 1.0 - 2.0 * (xx + yy);
  return r;
}
// Eigen's Quaternion::inverse() divides by the squared norm (Q4 of the survey);
// the stored quaternions are unit up to rounding, so R(q⁻¹) = R(q)ᵀ / |q|⁴·|q|²…
// we normalise explicitly to stay within rounding of the reference.
DEV Q4 normalized(Q4 q) {
  const double n = 1.0 / sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Q4 r; r.x = q.x * n; r.y = q.y * n; r.z = q.z * n; r.w = q.w * n; return r;
}
// ceres::AngleAxisToQuaternion (camera_cost_functor.h:122, accelerometer_cost_functor.h:115)
DEV Q4 angle_axis_to_quat(V3 a) {
  const double t2 = dot(a, a);
  Q4 q;
  if (t2 > 0.0) {
    const double t = sqrt(t2);
    const double h = 0.5 * t;
    double sh, ch;
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(h, &sh, &ch);
#else
    sh = sin(h); ch = cos(h);
#endif
    const double k = sh / t;
    q.w = ch; q.x = a.x * k; q.y = a.y * k; q.z = a.z * k;
  } else {
    q.w = 1.0; q.x = 0.5 * a.x; q.y = 0.5 * a.y; q.z = 0.5 * a.z;
  }
  return q;
}

// ---------------------------------------------------------------------------
// 3-lane forward dual for the Rodrigues pieces.
// ---------------------------------------------------------------------------
struct D3 { double v, d0, d1, d2; };
DEV D3 mkd(double v) { D3 r; r.v = v; r.d0 = r.d1 = r.d2 = 0.0; return r; }
DEV D3 operator+(D3 a, D3 b) { D3 r; r.v = a.v + b.v; r.d0 = a.d0 + b.d0; r.d1 = a.d1 + b.d1; r.d2 = a.d2 + b.d2; return r; }
DEV D3 operator-(D3 a, D3 b) { D3 r; r.v = a.v - b.v; r.d0 = a.d0 - b.d0; r.d1 = a.d1 - b.d1; r.d2 = a.d2 - b.d2; return r; }
DEV D3 operator-(D3 a) { D3 r; r.v = -a.v; r.d0 = -a.d0; r.d1 = -a.d1; r.d2 = -a.d2; return r; }
DEV D3 operator*(D3 a, D3 b) {
  D3 r; r.v = a.v * b.v; r.d0 = a.v * b.d0 + a.d0 * b.v; r.d1 = a.v * b.d1 + a.d1 * b.v; r.d2 = a.v * b.d2 + a.d2 * b.v; return r; }
DEV D3 operator*(double s, D3 a) { D3 r; r.v = s * a.v; r.d0 = s * a.d0; r.d1 = s * a.d1; r.d2 = s * a.d2; return r; }
DEV D3 operator/(D3 a, D3 b) {
  const double inv = 1.0 / b.v; D3 r; r.v = a.v * inv;
  r.d0 = (a.d0 - r.v * b.d0) * inv; r.d1 = (a.d1 - r.v * b.d1) * inv; r.d2 = (a.d2 - r.v * b.d2) * inv; return r; }
DEV D3 dsqrt(D3 a) { D3 r; r.v = sqrt(a.v); const double k = 0.5 / r.v; r.d0 = a.d0 * k; r.d1 = a.d1 * k; r.d2 = a.d2 * k; return r; }
DEV D3 dsin(D3 a) { D3 r; r.v = sin(a.v); const double c = cos(a.v); r.d0 = a.d0 * c; r.d1 = a.d1 * c; r.d2 = a.d2 * c; return r; }
DEV D3 dcos(D3 a) { D3 r; r.v = cos(a.v); const double s = -sin(a.v); r.d0 = a.d0 * s; r.d1 = a.d1 * s; r.d2 = a.d2 * s; return r; }
DEV double dsqrt(double a) { return sqrt(a); }
DEV double dsin(double a) { return sin(a); }
DEV double dcos(double a) { return cos(a); }
// sine and cosine of one angle share the argument reduction (on the device a double-precision sin or cos is some
// 150 instructions, and the IMU blocks sit on a single-lane latency chain)
DEV void dsincos(double a, double* s, double* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(a, s, c);
#else
  *s = sin(a); *c = cos(a);
#endif
}
DEV void dsincos(D3 a, D3* s, D3* c) {
  double sv, cv;
  dsincos(a.v, &sv, &cv);
  s->v = sv; s->d0 = a.d0 * cv; s->d1 = a.d1 * cv; s->d2 = a.d2 * cv;
  c->v = cv; c->d0 = -a.d0 * sv; c->d1 = -a.d1 * sv; c->d2 = -a.d2 * sv;
}
// One-direction forward dual: the IMU Jacobians deal the three φ-directions of an observation to three lanes, each
// lane differentiates along its own direction with the same operation sequence D3 uses per component.
struct D1 { double v, d; };
DEV D1 mk1(double v) { D1 r; r.v = v; r.d = 0.0; return r; }
DEV D1 operator+(D1 a, D1 b) { D1 r; r.v = a.v + b.v; r.d = a.d + b.d; return r; }
DEV D1 operator-(D1 a, D1 b) { D1 r; r.v = a.v - b.v; r.d = a.d - b.d; return r; }
DEV D1 operator-(D1 a) { D1 r; r.v = -a.v; r.d = -a.d; return r; }
DEV D1 operator*(D1 a, D1 b) { D1 r; r.v = a.v * b.v; r.d = a.v * b.d + a.d * b.v; return r; }
DEV D1 operator*(double s, D1 a) { D1 r; r.v = s * a.v; r.d = s * a.d; return r; }
DEV D1 operator/(D1 a, D1 b) { const double inv = 1.0 / b.v; D1 r; r.v = a.v * inv; r.d = (a.d - r.v * b.d) * inv; return r; }
DEV D1 dsqrt(D1 a) { D1 r; r.v = sqrt(a.v); const double k = 0.5 / r.v; r.d = a.d * k; return r; }
DEV void dsincos(D1 a, D1* s, D1* c) {
  double sv, cv;
  dsincos(a.v, &sv, &cv);
  s->v = sv; s->d = a.d * cv;
  c->v = cv; c->d = -a.d * sv;
}
// A value picked by the lane's place in a triple (jl = 0, 1, 2) WITHOUT control flow. Written as `jl == 0 ? a : (jl == 1 ?
// b : c)` the compiler threads the many choices on the same jl of an IMU block into exec-masked regions: some sixty
// branches (s_and_saveexec / s_cbranch) around register moves, a third of the block's instructions, every taken branch an
// instruction-fetch bubble on the longest single-wave chain of the evaluation. Here the choice is bit arithmetic on masks
// the optimiser cannot see through (an empty asm makes them opaque, or it would turn the and / or back into selects):
// two v_bfi_b32 per 32-bit half.
struct Lane3 { unsigned m0, m1; };      // all ones where jl == 0 / jl == 1
DEV Lane3 lane3(int jl) {
  Lane3 l; l.m0 = jl == 0 ? ~0u : 0u; l.m1 = jl == 1 ? ~0u : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(l.m0), "+v"(l.m1));
#endif
  return l;
}
DEV double pick3(const Lane3& l, double a, double b, double c) {
  unsigned long long ua, ub, uc;
  __builtin_memcpy(&ua, &a, 8); __builtin_memcpy(&ub, &b, 8); __builtin_memcpy(&uc, &c, 8);
  const unsigned alo = unsigned(ua), ahi = unsigned(ua >> 32), blo = unsigned(ub), bhi = unsigned(ub >> 32), clo = unsigned(uc), chi = unsigned(uc >> 32);
  const unsigned tlo = (blo & l.m1) | (clo & ~l.m1), thi = (bhi & l.m1) | (chi & ~l.m1);
  const unsigned rlo = (alo & l.m0) | (tlo & ~l.m0), rhi = (ahi & l.m0) | (thi & ~l.m0);
  const unsigned long long ur = (static_cast<unsigned long long>(rhi) << 32) | rlo;
  double r;
  __builtin_memcpy(&r, &ur, 8);
  return r;
}
DEV double val(double a) { return a; }
DEV double val(D3 a) { return a.v; }
DEV double val(D1 a) { return a.v; }
template <class T> DEV T lit(double v);
template <> DEV double lit<double>(double v) { return v; }
template <> DEV D3 lit<D3>(double v) { return mkd(v); }
template <> DEV D1 lit<D1>(double v) { return mk1(v); }

// geometry.h:35-50
template <class T> DEV T small_sin(T th) {
  const T t2 = th * th;
  return th * (lit<T>(1.0) - t2 * (lit<T>(1.0 / 6.0) + t2 * (lit<T>(1.0 / 120.0) - t2 * lit<T>(1.0 / 5040.0))));
}
template <class T> DEV T small_cos(T th) {
  const T t2 = th * th;
  return lit<T>(1.0) - t2 * (lit<T>(0.5) - t2 * (lit<T>(1.0 / 24.0) + t2 * (lit<T>(1.0 / 720.0) - t2 * lit<T>(1.0 / 40320.0))));
}

// Coefficients of ExpSO3Jacobian / ExpSO3Hessian in the unit-axis form the
// reference uses (geometry.h:137-161, 172-210):
//   J = I + a·K + b·K²,  K = skew(phi/theta), a = (1-cos)/theta, b = (theta-sin)/theta
//   H_i = c0·h_i·K + c1·G_i + c2·h_i·K² + c3·(G_i K + K G_i)
// Generic in T so the same code gives values (double) and φ-derivatives (D3).
template <class T> struct Rodrigues {
  T hx, hy, hz;      // unit axis
  T a, b;            // Jacobian coefficients
  T c0, c1, c2, c3;  // Hessian coefficients
  bool zero;
};
template <class T> DEV Rodrigues<T> rodrigues(T px, T py, T pz, bool want_hessian) {
  Rodrigues<T> R;
  const T t2 = px * px + py * py + pz * pz;
  R.zero = (val(t2) == 0.0);
  if (R.zero) {
    R.hx = R.hy = R.hz = lit<T>(0.0); R.a = R.b = lit<T>(0.0);
    R.c0 = R.c1 = R.c2 = R.c3 = lit<T>(0.0);
    return R;
  }
  const T th = dsqrt(t2);
  T st, ct;
  if (val(th) < 1e-7) { st = small_sin(th); ct = small_cos(th); }
  else { dsincos(th, &st, &ct); }
  const T it = lit<T>(1.0) / th;
  R.hx = it * px; R.hy = it * py; R.hz = it * pz;
  R.a = it * (lit<T>(1.0) - ct);
  R.b = it * (th - st);
  if (want_hessian) {
    const T it2 = it * it;
    R.c0 = ct - st * it;
    R.c1 = (lit<T>(1.0) - ct) * it2;
    R.c2 = lit<T>(3.0) * it2 * st - it * (ct - lit<T>(2.0));
    R.c3 = it2 * (th - st);
  } else {
    R.c0 = R.c1 = R.c2 = R.c3 = lit<T>(0.0);
  }
  return R;
}
// the plain-double coefficients of a dual evaluation (same phi): saves recomputing the trigonometry
DEV Rodrigues<double> rod_value(const Rodrigues<D3>& R) {
  Rodrigues<double> r;
  r.hx = R.hx.v; r.hy = R.hy.v; r.hz = R.hz.v; r.a = R.a.v; r.b = R.b.v;
  r.c0 = R.c0.v; r.c1 = R.c1.v; r.c2 = R.c2.v; r.c3 = R.c3.v; r.zero = R.zero;
  return r;
}
DEV Rodrigues<double> rod_value(const Rodrigues<D1>& R) {
  Rodrigues<double> r;
  r.hx = R.hx.v; r.hy = R.hy.v; r.hz = R.hz.v; r.a = R.a.v; r.b = R.b.v;
  r.c0 = R.c0.v; r.c1 = R.c1.v; r.c2 = R.c2.v; r.c3 = R.c3.v; r.zero = R.zero;
  return r;
}
// y = J(phi)·v, with K·v = h×v.
template <class T> DEV void rod_J_apply(const Rodrigues<T>& R, T vx, T vy, T vz, T* ox, T* oy, T* oz) {
  if (R.zero) { *ox = vx; *oy = vy; *oz = vz; return; }
  const T kx = R.hy * vz - R.hz * vy, ky = R.hz * vx - R.hx * vz, kz = R.hx * vy - R.hy * vx;         // K v
  const T k2x = R.hy * kz - R.hz * ky, k2y = R.hz * kx - R.hx * kz, k2z = R.hx * ky - R.hy * kx;     // K² v
  *ox = vx + R.a * kx + R.b * k2x; *oy = vy + R.a * ky + R.b * k2y; *oz = vz + R.a * kz + R.b * k2z;
}
// J as a matrix (double only).
DEV M3 rod_J_matrix(const Rodrigues<double>& R) {
  M3 J;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double ox, oy, oz;
    rod_J_apply<double>(R, c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0, &ox, &oy, &oz);
    J.m[0][c] = ox; J.m[1][c] = oy; J.m[2][c] = oz;
  }
  return J;
}
// w = H_i(phi)·v for i = 0..2 (geometry.h:204-209); out[i] = H_i v.
template <class T> DEV void rod_H_apply(const Rodrigues<T>& R, T vx, T vy, T vz, T out[3][3]) {
  if (R.zero) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i][0] = out[i][1] = out[i][2] = lit<T>(0.0);
    return;
  }
  const T h[3] = {R.hx, R.hy, R.hz};
  const T kx = R.hy * vz - R.hz * vy, ky = R.hz * vx - R.hx * vz, kz = R.hx * vy - R.hy * vx;      // K v
  const T k2x = R.hy * kz - R.hz * ky, k2y = R.hz * kx - R.hx * kz, k2z = R.hx * ky - R.hy * kx;  // K² v
  const T v[3] = {vx, vy, vz};
  const T kv[3] = {kx, ky, kz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // G_i u = e_i × u
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    T gv[3], gkv[3], kgv[3];
    gv[i] = lit<T>(0.0); gv[i1] = -v[i2]; gv[i2] = v[i1];            // e_i × v
    gkv[i] = lit<T>(0.0); gkv[i1] = -kv[i2]; gkv[i2] = kv[i1];       // G_i (K v)
    // K (G_i v) = h × gv
    kgv[0] = h[1] * gv[2] - h[2] * gv[1]; kgv[1] = h[2] * gv[0] - h[0] * gv[2]; kgv[2] = h[0] * gv[1] - h[1] * gv[0];
    const T s0 = R.c0 * h[i], s2 = R.c2 * h[i];
    out[i][0] = s0 * kx + R.c1 * gv[0] + s2 * k2x + R.c3 * (gkv[0] + kgv[0]);
    out[i][1] = s0 * ky + R.c1 * gv[1] + s2 * k2y + R.c3 * (gkv[1] + kgv[1]);
    out[i][2] = s0 * kz + R.c1 * gv[2] + s2 * k2z + R.c3 * (gkv[2] + kgv[2]);
  }
}

// ---------------------------------------------------------------------------
// Camera projection models with analytic derivatives.
// camera_models.h:104-141, 256-298, 419-462, 622-657, 739-781, 871-901, 984-1015.
// Outputs: pix[2]; D = d pix / d point (2×3); dK = d pix / d intrinsics (2×K).
// Returns false where the reference returns a non-OK status.
// ---------------------------------------------------------------------------
constexpr int kMaxIntr = 12;

template <int MODEL> struct CamK;
template <> struct CamK<1> { static constexpr int K = 8; };
template <> struct CamK<2> { static constexpr int K = 11; };
template <> struct CamK<3> { static constexpr int K = 7; };
template <> struct CamK<4> { static constexpr int K = 5; };
template <> struct CamK<5> { static constexpr int K = 4; };
template <> struct CamK<6> { static constexpr int K = 4; };
template <> struct CamK<7> { static constexpr int K = 5; };

template <int MODEL, bool JAC>
DEV bool project(const double* __restrict__ k, V3 P, double pix[2], double D[2][3], double dK[2][kMaxIntr]) {
  const double f = k[0], cx = k[1], cy = k[2];
  if constexpr (MODEL == 1 || MODEL == 2 || MODEL == 3 || MODEL == 5) {
    if (P.z <= 0.0) return false;
    const double iz = 1.0 / P.z;
    const double x = P.x * iz, y = P.y * iz;
    double dx, dy;                 // distorted normalised point
    double dxx, dxy, dyx, dyy;     // d(dx,dy)/d(x,y)
    if constexpr (MODEL == 1 || MODEL == 2) {
      const double k1 = k[3], k2 = k[4], p1 = k[5], p2 = k[6], k3 = k[7];
      const double r2 = x * x + y * y;
      double s, sp;  // s and ds/dr2
      const double num = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
      const double nump = k1 + r2 * (2.0 * k2 + 3.0 * r2 * k3);
      double den = 1.0, iden = 1.0;
      if constexpr (MODEL == 2) {
        den = 1.0 + r2 * (k[8] + r2 * (k[9] + r2 * k[10]));
        iden = 1.0 / den;
        s = num * iden;
        const double denp = k[8] + r2 * (2.0 * k[9] + 3.0 * r2 * k[10]);
        sp = (nump - s * denp) * iden;
      } else { s = num; sp = nump; }
      dx = x * s + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
      dy = y * s + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y);
      if constexpr (JAC) {
        dxx = s + 2.0 * x * x * sp + 2.0 * p1 * y + 6.0 * p2 * x;
        dxy = 2.0 * x * y * sp + 2.0 * p1 * x + 2.0 * p2 * y;
        dyx = 2.0 * x * y * sp + 2.0 * p2 * y + 2.0 * p1 * x;
        dyy = s + 2.0 * y * y * sp + 2.0 * p2 * x + 6.0 * p1 * y;
        const double r4 = r2 * r2, r6 = r4 * r2;
        dK[0][3] = f * x * r2 * iden; dK[1][3] = f * y * r2 * iden;
        dK[0][4] = f * x * r4 * iden; dK[1][4] = f * y * r4 * iden;
        dK[0][5] = f * 2.0 * x * y; dK[1][5] = f * (r2 + 2.0 * y * y);
        dK[0][6] = f * (r2 + 2.0 * x * x); dK[1][6] = f * 2.0 * x * y;
        dK[0][7] = f * x * r6 * iden; dK[1][7] = f * y * r6 * iden;
        if constexpr (MODEL == 2) {
          const double q = -s * iden;
          dK[0][8] = f * x * q * r2; dK[1][8] = f * y * q * r2;
          dK[0][9] = f * x * q * r4; dK[1][9] = f * y * q * r4;
          dK[0][10] = f * x * q * r6; dK[1][10] = f * y * q * r6;
        }
      }
    } else if constexpr (MODEL == 3) {
      const double k1 = k[3], k2 = k[4], k3 = k[5], k4 = k[6];
      const double r = sqrt(x * x + y * y);
      double s, dsdr_over_r;  // ds/dr / r, so that d s/dx = dsdr_over_r * x
      double sk[4];
      if (r < 1e-9) {
        const double r2 = r * r;
        s = 1.0 + r2 * (k1 - (1.0 / 3.0) + r2 * (-k1 + k2 + 0.2));
        dsdr_over_r = 2.0 * ((k1 - (1.0 / 3.0)) + 2.0 * r2 * (-k1 + k2 + 0.2));
        sk[0] = r2 - r2 * r2; sk[1] = r2 * r2; sk[2] = 0.0; sk[3] = 0.0;
      } else {
        const double th = atan(r);
        const double t2 = th * th;
        const double thd = th * (1.0 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4))));
        const double ir = 1.0 / r;
        s = thd * ir;
        const double dthd = 1.0 + t2 * (3.0 * k1 + t2 * (5.0 * k2 + t2 * (7.0 * k3 + t2 * 9.0 * k4)));
        const double dsdr = (dthd / (1.0 + r * r) - s) * ir;
        dsdr_over_r = dsdr * ir;
        const double t3 = t2 * th;
        sk[0] = t3 * ir; sk[1] = t3 * t2 * ir; sk[2] = t3 * t2 * t2 * ir; sk[3] = t3 * t2 * t2 * t2 * ir;
      }
      dx = x * s; dy = y * s;
      if constexpr (JAC) {
        dxx = s + x * x * dsdr_over_r; dxy = x * y * dsdr_over_r; dyx = dxy; dyy = s + y * y * dsdr_over_r;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dK[0][3 + i] = f * x * sk[i]; dK[1][3 + i] = f * y * sk[i]; }
      }
    } else {  // MODEL == 5, field of view
      const double w = k[3];
      const double r = sqrt(x * x + y * y);
      double s, dsdr_over_r = 0.0, dsdw = 0.0;
      if (w * w < 1e-5) {
        s = 1.0;
      } else {
        const double tt = 2.0 * tan(w * 0.5);
        const double dtt = 1.0 + 0.25 * tt * tt;
        if (r * r < 1e-5) {
          s = tt / w;
          dsdw = (dtt - s) / w;
        } else {
          const double arg = r * tt;
          const double at = atan(arg);
          const double irw = 1.0 / (r * w);
          s = at * irw;
          const double q = 1.0 / (1.0 + arg * arg);
          dsdr_over_r = (tt * q * irw - s / r) / r;
          dsdw = r * dtt * q * irw - s / w;
        }
      }
      dx = x * s; dy = y * s;
      if constexpr (JAC) {
        dxx = s + x * x * dsdr_over_r; dxy = x * y * dsdr_over_r; dyx = dxy; dyy = s + y * y * dsdr_over_r;
        dK[0][3] = f * x * dsdw; dK[1][3] = f * y * dsdw;
      }
    }
    pix[0] = dx * f + cx; pix[1] = dy * f + cy;
    if constexpr (JAC) {
      dK[0][0] = dx; dK[1][0] = dy; dK[0][1] = 1.0; dK[1][1] = 0.0; dK[0][2] = 0.0; dK[1][2] = 1.0;
      // d(x,y)/dP = [iz 0 -x iz; 0 iz -y iz]
      D[0][0] = f * dxx * iz; D[0][1] = f * dxy * iz; D[0][2] = -f * (dxx * x + dxy * y) * iz;
      D[1][0] = f * dyx * iz; D[1][1] = f * dyy * iz; D[1][2] = -f * (dyx * x + dyy * y) * iz;
    }
    return true;
  } else {
    // sphere-type models: pix = f * s * (X, Y) + c, s = 1/den
    double den, dden[3], dpar[2] = {0.0, 0.0};  // d den / dP, d den / d(k[3], k[4])
    if constexpr (MODEL == 4) {
      const double xi = k[3], al = k[4];
      const double w1 = al > 0.5 ? (1.0 - al) / al : al / (1.0 - al);
      const double num = w1 + xi;
      const double w2sq = num * num / (2.0 * w1 * xi + xi * xi + 1.0);
      const double r2 = dot(P, P);
      if (P.z * P.z <= -w2sq * r2) return false;
      const double r = sqrt(r2);
      const double d = sqrt(r2 * (1.0 + xi * xi) + 2.0 * xi * r * P.z);
      den = al * d + (1.0 - al) * (xi * r + P.z);
      if constexpr (JAC) {
        const double ir = 1.0 / r, id = 1.0 / d;
        const double dd[3] = {((1.0 + xi * xi) * P.x + xi * P.z * P.x * ir) * id,
                              ((1.0 + xi * xi) * P.y + xi * P.z * P.y * ir) * id,
                              ((1.0 + xi * xi) * P.z + xi * (P.z * P.z * ir + r)) * id};
        dden[0] = al * dd[0] + (1.0 - al) * xi * P.x * ir;
        dden[1] = al * dd[1] + (1.0 - al) * xi * P.y * ir;
        dden[2] = al * dd[2] + (1.0 - al) * (xi * P.z * ir + 1.0);
        dpar[0] = al * (xi * r2 + r * P.z) * id + (1.0 - al) * r;
        dpar[1] = d - (xi * r + P.z);
      }
    } else if constexpr (MODEL == 6) {
      const double al = k[3];
      const double w = al > 0.5 ? (1.0 - al) / al : al / (1.0 - al);
      const double d = sqrt(dot(P, P));
      if (P.z <= -w * d) return false;
      den = al * d + (1.0 - al) * P.z;
      if constexpr (JAC) {
        const double id = 1.0 / d;
        dden[0] = al * P.x * id; dden[1] = al * P.y * id; dden[2] = al * P.z * id + (1.0 - al);
        dpar[0] = d - P.z;
      }
    } else {  // MODEL == 7 (Q5: norm(), not squaredNorm())
      const double al = k[3], be = k[4];
      const double rho = sqrt(P.x * P.x + P.y * P.y);
      const double d = sqrt(be * rho + P.z * P.z);
      const double w = al > 0.5 ? (1.0 - al) / al : al / (1.0 - al);
      if (P.z <= -w * d) return false;
      den = al * d + (1.0 - al) * P.z;
      if constexpr (JAC) {
        const double id = 1.0 / d;
        const double irho = rho > 0.0 ? 1.0 / rho : 0.0;
        dden[0] = al * 0.5 * be * P.x * irho * id; dden[1] = al * 0.5 * be * P.y * irho * id;
        dden[2] = al * P.z * id + (1.0 - al);
        dpar[0] = d - P.z;
        dpar[1] = al * 0.5 * rho * id;
      }
    }
    const double s = 1.0 / den;
    pix[0] = P.x * s * f + cx; pix[1] = P.y * s * f + cy;
    if constexpr (JAC) {
      const double s2 = s * s;
      dK[0][0] = P.x * s; dK[1][0] = P.y * s; dK[0][1] = 1.0; dK[1][1] = 0.0; dK[0][2] = 0.0; dK[1][2] = 1.0;
      dK[0][3] = -f * P.x * s2 * dpar[0]; dK[1][3] = -f * P.y * s2 * dpar[0];
      if constexpr (MODEL != 6) { dK[0][4] = -f * P.x * s2 * dpar[1]; dK[1][4] = -f * P.y * s2 * dpar[1]; }
      D[0][0] = f * (s - P.x * s2 * dden[0]); D[0][1] = -f * P.x * s2 * dden[1]; D[0][2] = -f * P.x * s2 * dden[2];
      D[1][0] = -f * P.y * s2 * dden[0]; D[1][1] = f * (s - P.y * s2 * dden[1]); D[1][2] = -f * P.y * s2 * dden[2];
    }
    return true;
  }
}

// ---------------------------------------------------------------------------
// IMU intrinsic models (gyroscope_models.h:82-87,130-142,208-235; the
// accelerometer models are identical). f = M(k, w); Mw = df/dw (3×3);
// dK = df/dk (3×K) dense.
// ---------------------------------------------------------------------------
DEV int imu_num_params(int model) { return model == 1 ? 1 : (model == 2 ? 4 : 12); }
template <bool JAC>
DEV void imu_project(int model, const double* __restrict__ k, V3 w, double f[3], double Mw[3][3], double dK[3][kMaxIntr]) {
  if (model == 1 || model == 2) {
    const double s = k[0];
    f[0] = s * w.x; f[1] = s * w.y; f[2] = s * w.z;
    if (model == 2) { f[0] += k[1]; f[1] += k[2]; f[2] += k[3]; }
    if constexpr (JAC) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Mw[i][j] = (i == j) ? s : 0.0;
      // (the callers pick their component of every 3-wide group of columns without control flow: the groups that hold
      //  a column of this model -- [0, 6) -- must be defined throughout)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) dK[i][j] = 0.0;
      dK[0][0] = w.x; dK[1][0] = w.y; dK[2][0] = w.z;
      if (model == 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) dK[i][1 + j] = (i == j) ? 1.0 : 0.0;
      }
    }
  } else {
    const double sx = k[0], sy = k[1], sz = k[2], a1 = k[3], a2 = k[4], a3 = k[5], a4 = k[6], a5 = k[7], a6 = k[8];
    const double ux = w.x + a1 * w.y + a2 * w.z, uy = w.y + a3 * w.x + a4 * w.z, uz = w.z + a5 * w.x + a6 * w.y;
    f[0] = k[9] + sx * ux; f[1] = k[10] + sy * uy; f[2] = k[11] + sz * uz;
    if constexpr (JAC) {
      Mw[0][0] = sx; Mw[0][1] = sx * a1; Mw[0][2] = sx * a2;
      Mw[1][0] = sy * a3; Mw[1][1] = sy; Mw[1][2] = sy * a4;
      Mw[2][0] = sz * a5; Mw[2][1] = sz * a6; Mw[2][2] = sz;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) dK[i][j] = 0.0;
      dK[0][0] = ux; dK[1][1] = uy; dK[2][2] = uz;
      dK[0][3] = sx * w.y; dK[0][4] = sx * w.z;
      dK[1][5] = sy * w.x; dK[1][6] = sy * w.z;
      dK[2][7] = sz * w.x; dK[2][8] = sz * w.y;
      dK[0][9] = 1.0; dK[1][10] = 1.0; dK[2][11] = 1.0;
    }
  }
}

// ---------------------------------------------------------------------------
// Spline weights U·M for derivatives 0..ND-1 (bspline.hpp:39-72). k <= 8.
// ---------------------------------------------------------------------------
constexpr int kMaxOrder = 8;
// KT > 0 fixes the order at compile time (the runtime k is then ignored); every loop is unrolled to kMaxOrder
// with an `i < k` guard, so W / U are never indexed dynamically (no scratch) and the summation order is the same.
template <int ND, int KT = 0>
DEV void spline_weights(int k_rt, double knot0, double knot1, const double* __restrict__ M, double t,
                        double W[ND][kMaxOrder]) {
  const int k = KT > 0 ? KT : k_rt;
  const double dt_inv = 1.0 / (knot1 - knot0);
  const double u = (t - knot0) * dt_inv;
  double up[kMaxOrder];
  up[0] = 1.0;
#pragma unroll
  for (int i = 1; i < kMaxOrder; ++i) up[i] = u * up[i - 1];
  double scale = 1.0;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    double U[kMaxOrder];
#pragma unroll
    for (int i = 0; i < kMaxOrder; ++i) {
      double coeff = 1.0;
      for (int j = i - d; j < i; ++j) coeff *= double(j + 1);
      U[i] = (i >= d && i < k) ? coeff * up[i >= d ? i - d : 0] * scale : 0.0;
    }
#pragma unroll
    for (int j = 0; j < kMaxOrder; ++j) {
      double s = 0.0;
      if (j < k) {
#pragma unroll
        for (int i = 0; i < kMaxOrder; ++i) if (i < k) s += U[i] * M[i * k + j];
      }
      W[d][j] = s;
    }
    scale *= dt_inv;
  }
}

// Loss functions (ceres HuberLoss / CauchyLoss, optimization_utils.h:31-47).
// Both have rho'' <= 0, so Ceres' corrector reduces to scaling residual and
// Jacobian by sqrt(rho'). Returns rho(s); *scale = sqrt(rho').
DEV double loss_eval(int loss, double a, double s, double* scale) {
  if (loss == 1) {
    const double b = a * a;
    if (s > b) {
      const double r = sqrt(s);
      const double rho1 = fmax(2.2250738585072014e-308, a / r);
      *scale = sqrt(rho1);
      return 2.0 * a * r - b;
    }
    *scale = 1.0;
    return s;
  }
  if (loss == 2) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double rho1 = fmax(2.2250738585072014e-308, 1.0 / sum);
    *scale = sqrt(rho1);
    return b * log(sum);
  }
  *scale = 1.0;
  return s;
}

}  // namespace cal
