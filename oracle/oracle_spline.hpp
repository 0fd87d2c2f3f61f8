// oracle_spline.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle_math.hpp).
// Restatement of calico/bspline.{h,hpp} (general-matrix B-spline, K. Qin) and
// of the trajectory helpers in calico/trajectory.{h,cpp}.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <vector>

#include "oracle_math.hpp"

namespace oracle {

struct BSpline6 {
  int order = 0, degree = 0;
  double knot_frequency = 0;
  std::vector<double> knots, valid_knots;
  std::vector<std::vector<double>> Mi;          // per valid segment, k×k row-major
  std::vector<std::array<double, 6>> ctrl;

  // bspline.hpp:163-180
  void ComputeKnotVector(double t_front, double t_back) {
    const double duration = t_back - t_front;
    const double dt = 1.0 / knot_frequency;
    const int num_valid_knots = 1 + int(std::ceil(duration * knot_frequency));
    const int num_knots = num_valid_knots + 2 * degree;
    knots.assign(num_knots, 0.0);
    valid_knots.assign(num_valid_knots, 0.0);
    for (int i = -degree; i < num_knots - degree; ++i) {
      const double knot_value = t_front + dt * i;
      knots[i + degree] = knot_value;
      if (i > -1 && i < num_valid_knots) valid_knots[i] = knot_value;
    }
  }
  // bspline.hpp:226-244
  double d_0(int k, int i, int j) const {
    const double den = knots[j + k - 1] - knots[j];
    if (den <= 0.0) return 0.0;
    return (knots[i] - knots[j]) / den;
  }
  double d_1(int k, int i, int j) const {
    const double den = knots[j + k - 1] - knots[j];
    if (den <= 0.0) return 0.0;
    return (knots[i + 1] - knots[i]) / den;
  }
  // bspline.hpp:191-224: M_k = [M_{k-1};0]·A + [0;M_{k-1}]·B ; returns k×k row-major
  std::vector<double> M(int k, int i) const {
    if (k == 1) return std::vector<double>(1, double(k));
    const std::vector<double> Mkm1 = M(k - 1, i);
    const int nr = k - 1, nc = k - 1;
    std::vector<double> M1(k * nc, 0.0), M2(k * nc, 0.0);
    for (int r = 0; r < nr; ++r) for (int c = 0; c < nc; ++c) {
      M1[r * nc + c] = Mkm1[r * nc + c];
      M2[(r + 1) * nc + c] = Mkm1[r * nc + c];
    }
    std::vector<double> A((k - 1) * k, 0.0), B((k - 1) * k, 0.0);
    for (int index = 0; index < k - 1; ++index) {
      const int j = i - k + 2 + index;
      const double d0 = d_0(k, i, j), d1 = d_1(k, i, j);
      A[index * k + index] = 1.0 - d0; A[index * k + index + 1] = d0;
      B[index * k + index] = -d1;      B[index * k + index + 1] = d1;
    }
    std::vector<double> Mk(k * k, 0.0);
    for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) {
      double s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < k - 1; ++q) { s1 += M1[r * nc + q] * A[q * k + c]; s2 += M2[r * nc + q] * B[q * k + c]; }
      Mk[r * k + c] = s1 + s2;
    }
    return Mk;
  }
  // bspline.hpp:182-189
  void ComputeBasisMatrices() {
    const int nseg = int(valid_knots.size()) - 1;
    Mi.resize(nseg);
    for (int i = 0; i < nseg; ++i) Mi[i] = M(order, i + degree);
  }
  // bspline.hpp:138-150
  int GetSplineIndex(double t) const {
    int idx = -1;
    if (t == valid_knots.back()) idx = int(valid_knots.size()) - 2;
    else if (t < valid_knots.back()) {
      auto it = std::upper_bound(valid_knots.begin(), valid_knots.end(), t);
      idx = int(it - valid_knots.begin()) - 1;
    }
    return idx;
  }
  int GetKnotIndexFromSplineIndex(int i) const { return i + degree; }  // bspline.hpp:157-161

  // bspline.hpp:246-297. Dense normal equations; the reference solves
  // XtX with colPivHouseholderQr — here Gaussian elimination with partial
  // pivoting (initialisation only; not on the LM path).
  void FitSpline(const std::vector<double>& time, const std::vector<std::array<double, 6>>& data) {
    const int num_data = int(time.size());
    const int ncp = int(knots.size()) - order;
    std::vector<double> X(size_t(num_data) * ncp, 0.0);
    for (int j = 0; j < num_data; ++j) {
      const double t = time[j];
      int si = -1;
      if (t == valid_knots.back()) si = int(Mi.size()) - 1;
      else if (t == valid_knots.front()) si = 0;
      else if (t < valid_knots.back()) {
        auto it = std::upper_bound(valid_knots.begin(), valid_knots.end(), t);
        si = int(it - valid_knots.begin()) - 1;
      }
      const int ki = GetKnotIndexFromSplineIndex(si);
      const double ti = knots[ki], tii = knots[ki + 1];
      std::vector<double> U(order, 1.0);
      const double u = (t - ti) / (tii - ti);
      for (int i = 1; i < order; ++i) U[i] = u * U[i - 1];
      for (int c = 0; c < order; ++c) {
        double s = 0.0;
        for (int r = 0; r < order; ++r) s += U[r] * Mi[si][r * order + c];
        X[size_t(j) * ncp + si + c] = s;
      }
    }
    std::vector<double> XtX(size_t(ncp) * ncp, 0.0), Xtd(size_t(ncp) * 6, 0.0);
    for (int j = 0; j < num_data; ++j) {
      const double* row = &X[size_t(j) * ncp];
      int lo = 0; while (lo < ncp && row[lo] == 0.0) ++lo;
      int hi = ncp; while (hi > lo && row[hi - 1] == 0.0) --hi;
      for (int a = lo; a < hi; ++a) {
        for (int b = lo; b < hi; ++b) XtX[size_t(a) * ncp + b] += row[a] * row[b];
        for (int c = 0; c < 6; ++c) Xtd[size_t(a) * 6 + c] += row[a] * data[j][c];
      }
    }
    // Solve XtX · C = Xtd by column-pivoted Householder QR, as the reference
    // does (bspline.hpp:290-293, [Eigen] ColPivHouseholderQR). The fixture's
    // design matrix is rank deficient at the trajectory end (fewer samples
    // than control points there), where an untruncated solve is defined by
    // roundoff only; this restatement truncates at [Eigen] rank()'s threshold
    // (eps * n * max|R_ii|) and zeroes the dependent control points.
    const int n = ncp;
    std::vector<double>& A = XtX;
    std::vector<int> perm(n); for (int i = 0; i < n; ++i) perm[i] = i;
    std::vector<double> beta(n, 0.0);
    std::vector<double> cn(n);
    for (int c = 0; c < n; ++c) { double q = 0; for (int r = 0; r < n; ++r) q += A[size_t(r) * n + c] * A[size_t(r) * n + c]; cn[c] = q; }
    double maxpivot = 0.0;
    std::vector<double> v(n);
    for (int k = 0; k < n; ++k) {
      int p = k; double best = -1;
      for (int c = k; c < n; ++c) {
        double q = 0; for (int r = k; r < n; ++r) q += A[size_t(r) * n + c] * A[size_t(r) * n + c];
        cn[c] = q; if (q > best) { best = q; p = c; }
      }
      if (p != k) {
        for (int r = 0; r < n; ++r) std::swap(A[size_t(r) * n + k], A[size_t(r) * n + p]);
        std::swap(perm[k], perm[p]);
      }
      // Householder for column k, rows k..n-1
      double normx = std::sqrt(best);
      const double x0 = A[size_t(k) * n + k];
      double alpha = (x0 >= 0 ? -normx : normx);
      if (normx == 0.0) { beta[k] = 0; continue; }
      for (int r = k; r < n; ++r) v[r] = A[size_t(r) * n + k];
      v[k] -= alpha;
      double vtv = 0; for (int r = k; r < n; ++r) vtv += v[r] * v[r];
      beta[k] = vtv > 0 ? 2.0 / vtv : 0.0;
      for (int c = k; c < n; ++c) {
        double d = 0; for (int r = k; r < n; ++r) d += v[r] * A[size_t(r) * n + c];
        d *= beta[k];
        for (int r = k; r < n; ++r) A[size_t(r) * n + c] -= d * v[r];
      }
      for (int c = 0; c < 6; ++c) {
        double d = 0; for (int r = k; r < n; ++r) d += v[r] * Xtd[size_t(r) * 6 + c];
        d *= beta[k];
        for (int r = k; r < n; ++r) Xtd[size_t(r) * 6 + c] -= d * v[r];
      }
      maxpivot = std::max(maxpivot, std::fabs(A[size_t(k) * n + k]));
    }
    const double thr = maxpivot * 2.220446049250313e-16 * n;
    int rank = 0;
    for (int k = 0; k < n; ++k) if (std::fabs(A[size_t(k) * n + k]) > thr) ++rank; else break;
    std::vector<std::array<double, 6>> y(n);
    for (int r = n - 1; r >= 0; --r) {
      for (int q = 0; q < 6; ++q) {
        if (r >= rank) { y[r][q] = 0.0; continue; }
        double sacc = Xtd[size_t(r) * 6 + q];
        for (int c = r + 1; c < rank; ++c) sacc -= A[size_t(r) * n + c] * y[c][q];
        y[r][q] = sacc / A[size_t(r) * n + r];
      }
    }
    ctrl.assign(ncp, {});
    for (int r = 0; r < n; ++r) ctrl[perm[r]] = y[r];
  }
  // bspline.hpp:19-37 (time must be sorted: trajectory.cpp:24 sorts).
  bool FitToData(const std::vector<double>& time, const std::vector<std::array<double, 6>>& data, int spline_order,
                 double knot_freq) {
    if (time.empty() || data.empty() || time.size() != data.size() || spline_order < 2 || knot_freq <= 0) return false;
    order = spline_order; degree = order - 1; knot_frequency = knot_freq;
    ComputeKnotVector(time.front(), time.back());
    ComputeBasisMatrices();
    FitSpline(time, data);
    return true;
  }
  // bspline.hpp:74-100 (one time). Returns false where the reference errors.
  bool Interpolate(double t, int derivative, double out[6]) const {
    if (derivative < 0 || derivative > degree) return false;
    if (t < valid_knots.front() || t > valid_knots.back()) return false;
    const int si = GetSplineIndex(t);
    const int ki = GetKnotIndexFromSplineIndex(si);
    const double* cp[16];
    for (int j = 0; j < order; ++j) cp[j] = ctrl[si + j].data();
    SplineEvaluate<double>(cp, order, knots[ki], knots[ki + 1], Mi[si].data(), t, derivative, out);
    return true;
  }
};

// trajectory.cpp:81-93
inline void UnwrapPhaseLogMap(std::vector<V3<double>>& phi) {
  for (size_t i = 1; i < phi.size(); ++i) {
    const V3<double>& v1 = phi[i];
    const double theta = std::sqrt(squared_norm(v1));
    if (theta == 0) continue;
    const V3<double>& v0 = phi[i - 1];
    const double k = std::round((dot(v1, v0) - theta * theta) / (2.0 * M_PI * theta));
    phi[i] = (1.0 + 2.0 * M_PI * k / theta) * phi[i];
  }
}

// [Eigen] AngleAxisd(Quaterniond): angle = 2 atan2(|vec|, |w|), axis sign-adjusted.
inline V3<double> QuaternionToAngleAxisVector(const Quat<double>& q) {
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 2.2250738585072014e-308) n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);  // stableNorm no-op
  if (n != 0.0) {
    const double angle = 2.0 * std::atan2(n, std::fabs(q.w));
    const double s = (q.w < 0 ? -1.0 : 1.0) / n;
    return V3<double>(q.x * s * angle, q.y * s * angle, q.z * s * angle);
  }
  return V3<double>(0, 0, 0);  // angle 0, axis (1,0,0)
}

}  // namespace oracle
