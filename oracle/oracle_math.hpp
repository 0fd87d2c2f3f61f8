// oracle_math.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the arithmetic on the hot path of yangjames/Calico
// (BatchOptimizer::Optimize). Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build or call anything under oracle/.
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference). Arithmetic that lives in third-party code absent from the
// reference tree (Ceres Solver: Jet, AngleAxisToQuaternion, loss functions,
// corrector, EigenQuaternionManifold, trust-region LM; Eigen: quaternion
// product / inverse / vector rotation) is restated from its published
// algorithm and marked [Ceres] / [Eigen].
//
// Parity status: the reference cannot be built here (Eigen, Ceres, abseil
// absent), so this oracle is pinned only against the known answers the
// reference's own tests hold (see tests/test_oracle_*.py). Per-iteration LM
// behaviour, loss/corrector values and Jacobian values are "parity unpinned"
// by the reference; they are self-verified (dual numbers vs finite
// differences).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------
// Forward-mode dual number, N derivative lanes. [Ceres] ceres::Jet<double,N>.
// ---------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
  Dual(double s) : v(s) { for (int i = 0; i < N; ++i) d[i] = 0; }  // NOLINT
};

template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) {
  Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  // [Ceres] jet.h: a/b = (a.v/b.v, (a.d - a.v/b.v * b.d) / b.v)
  Dual<N> r; const double inv = 1.0 / b.v; r.v = a.v * inv;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double s) { Dual<N> r = a; r.v += s; return r; }
template <int N> inline Dual<N> operator+(double s, const Dual<N>& a) { Dual<N> r = a; r.v += s; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double s) { Dual<N> r = a; r.v -= s; return r; }
template <int N> inline Dual<N> operator-(double s, const Dual<N>& a) { return Dual<N>(s) - a; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double s) {
  Dual<N> r; r.v = a.v * s; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> inline Dual<N> operator*(double s, const Dual<N>& a) { return a * s; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double s) { return a * (1.0 / s); }
template <int N> inline Dual<N> operator/(double s, const Dual<N>& a) { return Dual<N>(s) / a; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> inline Dual<N>& operator/=(Dual<N>& a, const Dual<N>& b) { a = a / b; return a; }
// Comparisons act on the value part, as ceres::Jet does.
template <int N> inline bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
template <int N> inline bool operator<=(const Dual<N>& a, const Dual<N>& b) { return a.v <= b.v; }
template <int N> inline bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
template <int N> inline bool operator>=(const Dual<N>& a, const Dual<N>& b) { return a.v >= b.v; }
template <int N> inline bool operator==(const Dual<N>& a, const Dual<N>& b) { return a.v == b.v; }
template <int N> inline bool operator<(const Dual<N>& a, double b) { return a.v < b; }
template <int N> inline bool operator<=(const Dual<N>& a, double b) { return a.v <= b; }
template <int N> inline bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> inline bool operator==(const Dual<N>& a, double b) { return a.v == b; }

template <int N> inline Dual<N> sqrt(const Dual<N>& a) {
  Dual<N> r; r.v = std::sqrt(a.v); const double k = 1.0 / (2.0 * r.v);
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> inline Dual<N> sin(const Dual<N>& a) {
  Dual<N> r; r.v = std::sin(a.v); const double c = std::cos(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& a) {
  Dual<N> r; r.v = std::cos(a.v); const double s = -std::sin(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> inline Dual<N> tan(const Dual<N>& a) {
  Dual<N> r; r.v = std::tan(a.v); const double k = 1.0 + r.v * r.v;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
template <int N> inline Dual<N> atan(const Dual<N>& a) {
  Dual<N> r; r.v = std::atan(a.v); const double k = 1.0 / (1.0 + a.v * a.v);
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k; return r; }
inline double sqrt(double a) { return std::sqrt(a); }
inline double sin(double a) { return std::sin(a); }
inline double cos(double a) { return std::cos(a); }
inline double tan(double a) { return std::tan(a); }
inline double atan(double a) { return std::atan(a); }

inline double value_of(double a) { return a; }
template <int N> inline double value_of(const Dual<N>& a) { return a.v; }

// ---------------------------------------------------------------------------
// Small fixed-size algebra ([Eigen] semantics where it matters).
// ---------------------------------------------------------------------------
template <class T> struct V3 {
  T x, y, z;
  V3() : x(T(0.0)), y(T(0.0)), z(T(0.0)) {}
  V3(T a, T b, T c) : x(a), y(b), z(c) {}
  T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> inline V3<T> operator-(const V3<T>& a) { return V3<T>(-a.x, -a.y, -a.z); }
template <class T> inline V3<T> operator*(const T& s, const V3<T>& a) { return V3<T>(s * a.x, s * a.y, s * a.z); }
template <class T> inline V3<T> operator*(const V3<T>& a, const T& s) { return V3<T>(a.x * s, a.y * s, a.z * s); }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return V3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> inline T squared_norm(const V3<T>& a) { return dot(a, a); }

template <class T> struct M3 {
  T m[3][3];
  M3() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = T(0.0); }
  static M3 Identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1.0); return r; }
};
template <class T> inline M3<T> operator+(const M3<T>& a, const M3<T>& b) {
  M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
template <class T> inline M3<T> operator*(const M3<T>& a, const M3<T>& b) {
  M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    T s = a.m[i][0] * b.m[0][j]; s = s + a.m[i][1] * b.m[1][j]; s = s + a.m[i][2] * b.m[2][j]; r.m[i][j] = s; }
  return r; }
template <class T> inline M3<T> operator*(const T& s, const M3<T>& a) {
  M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j]; return r; }
template <class T> inline V3<T> operator*(const M3<T>& a, const V3<T>& v) {
  V3<T> r; for (int i = 0; i < 3; ++i) { T s = a.m[i][0] * v.x; s = s + a.m[i][1] * v.y; s = s + a.m[i][2] * v.z; r[i] = s; }
  return r; }
template <class T> inline M3<T> operator-(const M3<T>& a) {
  M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = -a.m[i][j]; return r; }

// [Eigen] Quaternion, storage coeffs() = (x,y,z,w); ctor order (w,x,y,z).
template <class T> struct Quat {
  T w, x, y, z;
  Quat() : w(T(1.0)), x(T(0.0)), y(T(0.0)), z(T(0.0)) {}
  Quat(T w_, T x_, T y_, T z_) : w(w_), x(x_), y(y_), z(z_) {}
  // From an Eigen coeffs() block (x,y,z,w): Eigen::Map<const Quaternion<T>>.
  static Quat FromCoeffs(const T* c) { return Quat(c[3], c[0], c[1], c[2]); }
  V3<T> vec() const { return V3<T>(x, y, z); }
  // [Eigen] QuaternionBase::inverse(): conjugate / squaredNorm (Q4).
  Quat inverse() const {
    const T n2 = w * w + x * x + y * y + z * z;
    if (n2 > T(0.0)) return Quat(w / n2, -x / n2, -y / n2, -z / n2);
    return Quat(T(0.0), T(0.0), T(0.0), T(0.0));
  }
  Quat conjugate() const { return Quat(w, -x, -y, -z); }
};
// [Eigen] quaternion product.
template <class T> inline Quat<T> operator*(const Quat<T>& a, const Quat<T>& b) {
  return Quat<T>(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                 a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                 a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
// [Eigen] QuaternionBase::_transformVector: v + w*2(u×v) + u×(2(u×v)).
template <class T> inline V3<T> operator*(const Quat<T>& q, const V3<T>& v) {
  V3<T> uv = cross(q.vec(), v);
  uv = uv + uv;
  return v + q.w * uv + cross(q.vec(), uv);
}

// [Ceres] rotation.h AngleAxisToQuaternion; output order w,x,y,z.
template <class T> inline Quat<T> AngleAxisToQuaternion(const V3<T>& a) {
  const T theta_squared = a.x * a.x + a.y * a.y + a.z * a.z;
  if (theta_squared > T(0.0)) {
    const T theta = sqrt(theta_squared);
    const T half_theta = theta * T(0.5);
    const T k = sin(half_theta) / theta;
    return Quat<T>(cos(half_theta), a.x * k, a.y * k, a.z * k);
  }
  const T k(0.5);
  return Quat<T>(T(1.0), a.x * k, a.y * k, a.z * k);
}

// ---------------------------------------------------------------------------
// calico/geometry.h
// ---------------------------------------------------------------------------
// geometry.h:11-22
template <class T> inline M3<T> Skew(const V3<T>& v) {
  M3<T> V;
  V.m[0][1] = -v.z; V.m[1][0] = v.z;
  V.m[0][2] = v.y;  V.m[2][0] = -v.y;
  V.m[1][2] = -v.x; V.m[2][1] = v.x;
  return V;
}
// geometry.h:25-32
template <class T> inline V3<T> iSkew(const M3<T>& V) {
  return T(0.5) * V3<T>(V.m[2][1] - V.m[1][2], V.m[0][2] - V.m[2][0], V.m[1][0] - V.m[0][1]);
}
// geometry.h:35-41
template <class T> inline T SmallAngleSin(const T theta) {
  const T theta_sq = theta * theta;
  return theta * (T(1.0) - theta_sq * (T(1.0 / 6.0) + theta_sq * (T(1.0 / 120.0) - theta_sq * T(1.0 / 5040.0))));
}
// geometry.h:44-50
template <class T> inline T SmallAngleCos(const T theta) {
  const T theta_sq = theta * theta;
  return T(1.0) - theta_sq * (T(0.5) - theta_sq * (T(1.0 / 24.0) + theta_sq * (T(1.0 / 720.0) - theta_sq * T(1.0 / 40320.0))));
}
// geometry.h:53-74
template <class T> inline M3<T> ExpSO3(const V3<T>& phi) {
  const T theta = sqrt(squared_norm(phi));
  if (theta == T(0.0)) return M3<T>::Identity();
  T sin_theta, one_m_cos_theta;
  if (theta < T(1e-7)) { sin_theta = SmallAngleSin(theta); one_m_cos_theta = T(1.0) - SmallAngleCos(theta); }
  else { sin_theta = sin(theta); one_m_cos_theta = T(1.0) - cos(theta); }
  const V3<T> phi_hat(phi.x / theta, phi.y / theta, phi.z / theta);
  const M3<T> Phi = Skew(phi_hat);
  return M3<T>::Identity() + sin_theta * Phi + one_m_cos_theta * (Phi * Phi);
}
// geometry.h:78-124 (double only; test support)
inline V3<double> LnSO3(const M3<double>& R) {
  const double kInvSqrt2 = 1.0 / std::sqrt(2.0);
  const double kSmallAngle = 1e-7;
  const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (tr == 3.0) return V3<double>();
  V3<double> phi = iSkew(R);
  const double cos_theta = 0.5 * (tr - 1.0);
  const double sin_theta = std::sqrt(squared_norm(phi));
  if (cos_theta >= kInvSqrt2) {
    const double theta = std::asin(sin_theta);
    const double st = theta < kSmallAngle ? SmallAngleSin(theta) : std::sin(theta);
    phi = (theta / st) * phi;
  } else if (cos_theta > -kInvSqrt2) {
    const double theta = std::acos(cos_theta);
    const double st = theta < kSmallAngle ? SmallAngleSin(theta) : std::sin(theta);
    phi = (theta / st) * phi;
  } else {
    const V3<double> diag(R.m[0][0] - cos_theta, R.m[1][1] - cos_theta, R.m[2][2] - cos_theta);
    const double dx2 = diag.x * diag.x, dy2 = diag.y * diag.y, dz2 = diag.z * diag.z;
    V3<double> axis;
    if ((dx2 > dz2) && (dx2 > dy2)) {
      axis = V3<double>(diag.x, 0.5 * (R.m[0][1] + R.m[1][0]), 0.5 * (R.m[0][2] + R.m[2][0]));
    } else if (dy2 > dz2) {
      axis = V3<double>(0.5 * (R.m[1][0] + R.m[0][1]), diag.y, 0.5 * (R.m[1][2] + R.m[2][1]));
    } else {
      axis = V3<double>(0.5 * (R.m[2][0] + R.m[0][2]), 0.5 * (R.m[2][1] + R.m[1][2]), diag.z);
    }
    if (dot(phi, axis) < 0.0) axis = -axis;
    const double theta = M_PI - std::asin(sin_theta);
    const double n = std::sqrt(squared_norm(axis));
    phi = (theta / n) * axis;
  }
  return phi;
}
// geometry.h:137-161
template <class T> inline M3<T> ExpSO3Jacobian(const V3<T>& phi) {
  const T theta_sq = squared_norm(phi);
  M3<T> J = M3<T>::Identity();
  if (theta_sq == T(0.0)) return J;
  const T theta = sqrt(theta_sq);
  T one_m_cos_theta, sin_theta;
  if (theta < T(1e-7)) { sin_theta = SmallAngleSin(theta); one_m_cos_theta = T(1.0) - SmallAngleCos(theta); }
  else { sin_theta = sin(theta); one_m_cos_theta = T(1.0) - cos(theta); }
  const T inv_theta = T(1.0) / theta;
  const V3<T> phi_hat = inv_theta * phi;
  const M3<T> phi_hat_x = Skew(phi_hat);
  return J + inv_theta * (one_m_cos_theta * phi_hat_x + (theta - sin_theta) * (phi_hat_x * phi_hat_x));
}
// geometry.h:172-210
template <class T> inline void ExpSO3Hessian(const V3<T>& phi, M3<T> H[3]) {
  M3<T> G[3] = {Skew(V3<T>(T(1.0), T(0.0), T(0.0))), Skew(V3<T>(T(0.0), T(1.0), T(0.0))),
                Skew(V3<T>(T(0.0), T(0.0), T(1.0)))};
  for (int i = 0; i < 3; ++i) H[i] = M3<T>();
  const T theta_sq = squared_norm(phi);
  if (theta_sq == T(0.0)) return;
  const T theta = sqrt(theta_sq);
  T ct, st;
  if (theta < T(1e-7)) { ct = SmallAngleCos(theta); st = SmallAngleSin(theta); }
  else { ct = cos(theta); st = sin(theta); }
  const T inv_theta = T(1.0) / theta;
  const T inv_theta_sq = inv_theta * inv_theta;
  const V3<T> phi_hat = inv_theta * phi;
  const M3<T> phi_hat_x = Skew(phi_hat);
  const T c0 = ct - st * inv_theta;
  const T c1 = (T(1.0) - ct) * inv_theta_sq;
  const T c2 = T(3.0) * inv_theta_sq * st - inv_theta * (ct - T(2.0));
  const T c3 = inv_theta_sq * (theta - st);
  for (int i = 0; i < 3; ++i) {
    H[i] = (c0 * phi_hat[i]) * phi_hat_x + c1 * G[i] + (c2 * phi_hat[i]) * (phi_hat_x * phi_hat_x) +
           c3 * (G[i] * phi_hat_x + phi_hat_x * G[i]);
  }
}
// geometry.h:213-222
template <class T> inline M3<T> ExpSO3JacobianDot(const V3<T>& phi, const V3<T>& phi_dot) {
  M3<T> H[3];
  ExpSO3Hessian(phi, H);
  M3<T> Jdot;
  for (int i = 0; i < 3; ++i) {
    const V3<T> col = H[i] * phi_dot;
    Jdot.m[0][i] = col.x; Jdot.m[1][i] = col.y; Jdot.m[2][i] = col.z;
  }
  return Jdot;
}

// ---------------------------------------------------------------------------
// calico/bspline.hpp:39-72  BSpline<6,T>::Evaluate
// control points: k rows of 6; basis k×k row-major (double).
// ---------------------------------------------------------------------------
template <class T>
inline void SplineEvaluate(const T* const* ctrl, int k, double knot0_d, double knot1_d,
                           const double* basis, const T& stamp, int derivative, T out[6]) {
  const T knot0 = T(knot0_d), knot1 = T(knot1_d);
  const T dt = knot1 - knot0;
  const T dt_inv = T(1.0) / dt;
  const T u = (stamp - knot0) * dt_inv;
  T dnu_dtn = T(1.0);
  for (int j = 0; j < derivative; ++j) dnu_dtn = dnu_dtn * dt_inv;
  T U[16], coeffs[16];
  for (int i = 0; i < k; ++i) { coeffs[i] = (i < derivative) ? T(0.0) : T(1.0); U[i] = T(1.0); }
  for (int i = derivative; i < k; ++i) {
    T coeff = T(1.0);
    for (int j = i - derivative; j < i; ++j) coeff = coeff * T(double(j + 1));
    coeffs[i] = coeff;
    U[i] = (i > derivative) ? (u * U[i - 1]) : U[i];
  }
  for (int i = 0; i < k; ++i) U[i] = U[i] * coeffs[i] * dnu_dtn;
  // (U * M) * C, left to right as Eigen evaluates the product chain.
  T UM[16];
  for (int j = 0; j < k; ++j) {
    T s = T(0.0);
    for (int i = 0; i < k; ++i) s = s + U[i] * T(basis[i * k + j]);
    UM[j] = s;
  }
  for (int c = 0; c < 6; ++c) {
    T s = T(0.0);
    for (int j = 0; j < k; ++j) s = s + UM[j] * ctrl[j][c];
    out[c] = s;
  }
}

// ---------------------------------------------------------------------------
// calico/sensors/camera_models.h  ProjectPoint ×7. Returns false where the
// reference returns a non-OK status.
// ---------------------------------------------------------------------------
enum CameraModelId { kCamNone = 0, kOpenCv5 = 1, kOpenCv8 = 2, kKannalaBrandt = 3, kDoubleSphere = 4,
                     kFieldOfView = 5, kUnifiedCamera = 6, kExtendedUnifiedCamera = 7 };
inline int CameraNumParams(int model) {
  switch (model) { case kOpenCv5: return 8; case kOpenCv8: return 11; case kKannalaBrandt: return 7;
    case kDoubleSphere: return 5; case kFieldOfView: return 4; case kUnifiedCamera: return 4;
    case kExtendedUnifiedCamera: return 5; default: return -1; }
}
template <class T>
inline bool ProjectPoint(int model, const T* in, const V3<T>& p, T out[2]) {
  switch (model) {
    case kOpenCv5: {  // camera_models.h:104-141
      if (p.z <= T(0.0)) return false;
      const T &f = in[0], &cx = in[1], &cy = in[2], &k1 = in[3], &k2 = in[4], &p1 = in[5], &p2 = in[6], &k3 = in[7];
      const T x = p.x / p.z, y = p.y / p.z;
      const T r2 = x * x + y * y;
      const T s = T(1.0) + r2 * (k1 + r2 * (k2 + r2 * k3));
      T px = x * s, py = y * s;
      px = px + (T(2.0) * p1 * x * y + p2 * (r2 + T(2.0) * x * x));
      py = py + (T(2.0) * p2 * x * y + p1 * (r2 + T(2.0) * y * y));
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    case kOpenCv8: {  // camera_models.h:256-298
      if (p.z <= T(0.0)) return false;
      const T &f = in[0], &cx = in[1], &cy = in[2], &k1 = in[3], &k2 = in[4], &p1 = in[5], &p2 = in[6], &k3 = in[7],
              &k4 = in[8], &k5 = in[9], &k6 = in[10];
      const T x = p.x / p.z, y = p.y / p.z;
      const T r2 = x * x + y * y;
      const T s_num = T(1.0) + r2 * (k1 + r2 * (k2 + r2 * k3));
      const T s_den = T(1.0) + r2 * (k4 + r2 * (k5 + r2 * k6));
      const T s = s_num / s_den;
      T px = x * s, py = y * s;
      px = px + (T(2.0) * p1 * x * y + p2 * (r2 + T(2.0) * x * x));
      py = py + (T(2.0) * p2 * x * y + p1 * (r2 + T(2.0) * y * y));
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    case kKannalaBrandt: {  // camera_models.h:419-462
      if (p.z <= T(0.0)) return false;
      const T &f = in[0], &cx = in[1], &cy = in[2], &k1 = in[3], &k2 = in[4], &k3 = in[5], &k4 = in[6];
      const T x = p.x / p.z, y = p.y / p.z;
      const T r = sqrt(x * x + y * y);
      T s;
      if (r < T(1e-9)) {
        const T r2 = r * r;
        s = T(1.0) + r2 * (k1 - T(1.0 / 3.0) + r2 * (-k1 + k2 + 0.2));
      } else {
        const T theta = atan(r);
        const T theta2 = theta * theta;
        const T theta_d = theta * (T(1.0) + theta2 * (k1 + theta2 * (k2 + theta2 * (k3 + theta2 * k4))));
        s = theta_d / r;
      }
      T px = x * s, py = y * s;
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    case kDoubleSphere: {  // camera_models.h:622-657
      const T &xi = in[3], &alpha = in[4];
      const T w1 = alpha > T(0.5) ? (T(1.0) - alpha) / alpha : alpha / (T(1.0) - alpha);
      const T num = w1 + xi;
      const T w2_sq = num * num / (T(2.0) * w1 * xi + xi * xi + T(1.0));
      const T r2 = squared_norm(p);
      if (p.z * p.z <= -w2_sq * r2) return false;
      const T &f = in[0], &cx = in[1], &cy = in[2];
      const T r = sqrt(r2);
      const T d = sqrt(r2 * (T(1.0) + xi * xi) + T(2.0) * xi * r * p.z);
      const T s = T(1.0) / (alpha * d + (T(1.0) - alpha) * (xi * r + p.z));
      T px = p.x * s, py = p.y * s;
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    case kFieldOfView: {  // camera_models.h:739-781
      const T &f = in[0], &cx = in[1], &cy = in[2], &w = in[3];
      if (p.z <= T(0.0)) return false;
      const T x = p.x / p.z, y = p.y / p.z;
      const T r = sqrt(x * x + y * y);
      T s;
      if (w * w < 1e-5) {
        s = T(1.0);
      } else {
        const T tan_term = T(2.0) * tan(w * T(0.5));
        if (r * r < 1e-5) s = tan_term / w;
        else s = atan(r * tan_term) / (r * w);
      }
      T px = x * s, py = y * s;
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    case kUnifiedCamera: {  // camera_models.h:871-901
      const T& alpha = in[3];
      const T w = alpha > T(0.5) ? (T(1.0) - alpha) / alpha : alpha / (T(1.0) - alpha);
      const T d = sqrt(squared_norm(p));
      if (p.z <= -w * d) return false;
      const T &f = in[0], &cx = in[1], &cy = in[2];
      const T s = T(1.0) / (alpha * d + (T(1.0) - alpha) * p.z);
      T px = p.x * s, py = p.y * s;
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    case kExtendedUnifiedCamera: {  // camera_models.h:984-1015 (Q5: norm(), not squaredNorm())
      const T &alpha = in[3], &beta = in[4];
      const T d = sqrt(beta * sqrt(p.x * p.x + p.y * p.y) + p.z * p.z);
      const T w = alpha > T(0.5) ? (T(1.0) - alpha) / alpha : alpha / (T(1.0) - alpha);
      if (p.z <= -w * d) return false;
      const T &f = in[0], &cx = in[1], &cy = in[2];
      const T s = T(1.0) / (alpha * d + (T(1.0) - alpha) * p.z);
      T px = p.x * s, py = p.y * s;
      px = px * f; py = py * f;
      out[0] = px + cx; out[1] = py + cy;
      return true;
    }
    default: return false;
  }
}

// ---------------------------------------------------------------------------
// gyroscope_models.h:82-87,130-142,208-235 and accelerometer_models.h (same).
// ---------------------------------------------------------------------------
enum ImuModelId { kImuNone = 0, kImuScaleOnly = 1, kImuScaleAndBias = 2, kImuVectorNav = 3 };
inline int ImuNumParams(int model) {
  switch (model) { case kImuScaleOnly: return 1; case kImuScaleAndBias: return 4; case kImuVectorNav: return 12;
    default: return -1; }
}
template <class T>
inline bool ImuProject(int model, const T* in, const V3<T>& w, V3<T>* out) {
  switch (model) {
    case kImuScaleOnly: *out = in[0] * w; return true;
    case kImuScaleAndBias: *out = in[0] * w + V3<T>(in[1], in[2], in[3]); return true;
    case kImuVectorNav: {
      const T &sx = in[0], &sy = in[1], &sz = in[2], &a1 = in[3], &a2 = in[4], &a3 = in[5], &a4 = in[6], &a5 = in[7],
              &a6 = in[8], &bx = in[9], &by = in[10], &bz = in[11];
      out->x = bx + sx * (w.x + a1 * w.y + a2 * w.z);
      out->y = by + sy * (w.y + a3 * w.x + a4 * w.z);
      out->z = bz + sz * (w.z + a5 * w.x + a6 * w.y);
      return true;
    }
    default: return false;
  }
}

// Evaluation parameters frozen at functor construction from the RAW stamp
// (Q3): trajectory.cpp:63-79, camera_cost_functor.cpp:8-16.
struct EvalParams {
  int k;               // spline order
  double knot0, knot1;
  double stamp;        // raw measurement stamp
  const double* basis; // k×k row-major, the segment's own matrix
  double information;  // sigma > 0 ? 1/sigma : 1
};

// ---------------------------------------------------------------------------
// camera_cost_functor.h:71-147. params order: camera_cost_functor.h:12-32
//  [intrinsics, q_rig_cam(x,y,z,w), t_rig_cam, latency, model_point,
//   q_world_model(x,y,z,w), t_world_model, ctrl[0..k-1]]
// ---------------------------------------------------------------------------
template <class T>
inline bool CameraResidual(int model, const EvalParams& ep, const double pixel[2],
                           T const* const* params, T* residual) {
  const T* intrinsics = params[0];
  const Quat<T> q_sensorrig_camera = Quat<T>::FromCoeffs(params[1]);
  const V3<T> t_sensorrig_camera(params[2][0], params[2][1], params[2][2]);
  const T latency = params[3][0];
  const V3<T> t_model_point(params[4][0], params[4][1], params[4][2]);
  const Quat<T> q_world_model = Quat<T>::FromCoeffs(params[5]);
  const V3<T> t_world_model(params[6][0], params[6][1], params[6][2]);
  const T stamp = T(ep.stamp) - latency;
  T pose[6];
  SplineEvaluate<T>(params + 7, ep.k, ep.knot0, ep.knot1, ep.basis, stamp, 0, pose);
  const V3<T> phi_sensorrig_world(-pose[0], -pose[1], -pose[2]);
  const Quat<T> q_sensorrig_world = AngleAxisToQuaternion(phi_sensorrig_world);
  const V3<T> t_world_sensorrig(pose[3], pose[4], pose[5]);
  const Quat<T> q_camera_model = q_sensorrig_camera.inverse() * q_sensorrig_world * q_world_model;
  const V3<T> t_world_camera = t_world_sensorrig + q_sensorrig_world.inverse() * t_sensorrig_camera;
  const V3<T> t_model_camera = q_world_model.inverse() * (t_world_camera - t_world_model);
  const V3<T> t_camera_point = q_camera_model * (t_model_point - t_model_camera);
  T proj[2];
  if (!ProjectPoint<T>(model, intrinsics, t_camera_point, proj)) return false;
  residual[0] = (T(pixel[0]) - proj[0]) * T(ep.information);
  residual[1] = (T(pixel[1]) - proj[1]) * T(ep.information);
  return true;
}

// gyroscope_cost_functor.h:58-118. params: [intrinsics, q_rig_gyro, t_rig_gyro, latency, ctrl...]
template <class T>
inline bool GyroscopeResidual(int model, const EvalParams& ep, const double meas[3],
                              T const* const* params, T* residual) {
  const T* intrinsics = params[0];
  const Quat<T> q_sensorrig_gyroscope = Quat<T>::FromCoeffs(params[1]);
  const T latency = params[3][0];
  const T stamp = T(ep.stamp) - latency;
  T pose[6], pose_dot[6];
  SplineEvaluate<T>(params + 4, ep.k, ep.knot0, ep.knot1, ep.basis, stamp, 0, pose);
  SplineEvaluate<T>(params + 4, ep.k, ep.knot0, ep.knot1, ep.basis, stamp, 1, pose_dot);
  const V3<T> phi(-pose[0], -pose[1], -pose[2]);
  const V3<T> phi_dot(-pose_dot[0], -pose_dot[1], -pose_dot[2]);
  const M3<T> J = ExpSO3Jacobian(phi);
  const V3<T> omega_sensorrig_world = J * phi_dot;
  const V3<T> omega_gyroscope_world = -(q_sensorrig_gyroscope.inverse() * omega_sensorrig_world);
  V3<T> proj;
  if (!ImuProject<T>(model, intrinsics, omega_gyroscope_world, &proj)) return false;
  residual[0] = (T(meas[0]) - proj.x) * T(ep.information);
  residual[1] = (T(meas[1]) - proj.y) * T(ep.information);
  residual[2] = (T(meas[2]) - proj.z) * T(ep.information);
  return true;
}

// accelerometer_cost_functor.h:62-147. params: [intrinsics, q_rig_acc, t_rig_acc, latency, gravity, ctrl...]
template <class T>
inline bool AccelerometerResidual(int model, const EvalParams& ep, const double meas[3],
                                  T const* const* params, T* residual) {
  const T* intrinsics = params[0];
  const Quat<T> q_sensorrig_accelerometer = Quat<T>::FromCoeffs(params[1]);
  const V3<T> t_sensorrig_accelerometer(params[2][0], params[2][1], params[2][2]);
  const T latency = params[3][0];
  const V3<T> gravity(params[4][0], params[4][1], params[4][2]);
  const T stamp = T(ep.stamp) - latency;
  T pose[6], pose_dot[6], pose_ddot[6];
  SplineEvaluate<T>(params + 5, ep.k, ep.knot0, ep.knot1, ep.basis, stamp, 0, pose);
  SplineEvaluate<T>(params + 5, ep.k, ep.knot0, ep.knot1, ep.basis, stamp, 1, pose_dot);
  SplineEvaluate<T>(params + 5, ep.k, ep.knot0, ep.knot1, ep.basis, stamp, 2, pose_ddot);
  const V3<T> phi(-pose[0], -pose[1], -pose[2]);
  const V3<T> phi_dot(-pose_dot[0], -pose_dot[1], -pose_dot[2]);
  const V3<T> phi_ddot(-pose_ddot[0], -pose_ddot[1], -pose_ddot[2]);
  const V3<T> ddt_world_sensorrig(pose_ddot[3], pose_ddot[4], pose_ddot[5]);
  const Quat<T> q_sensorrig_world = AngleAxisToQuaternion(phi);
  const M3<T> J = ExpSO3Jacobian(phi);
  const M3<T> J_dot = ExpSO3JacobianDot(phi, phi_dot);
  const V3<T> omega = J * phi_dot;
  const V3<T> alpha = J_dot * phi_dot + J * phi_ddot;
  const M3<T> Alpha = -Skew(alpha);
  const M3<T> Omega = -Skew(omega);
  const V3<T> f = q_sensorrig_accelerometer.inverse() *
                  (q_sensorrig_world * (ddt_world_sensorrig - gravity) +
                   (Omega * Omega + Alpha) * t_sensorrig_accelerometer);
  V3<T> proj;
  if (!ImuProject<T>(model, intrinsics, f, &proj)) return false;
  residual[0] = (T(meas[0]) - proj.x) * T(ep.information);
  residual[1] = (T(meas[1]) - proj.y) * T(ep.information);
  residual[2] = (T(meas[2]) - proj.z) * T(ep.information);
  return true;
}

}  // namespace oracle
