// calico.hpp — header-only C++ host side above the C ABI of libcalico_hip.so.
//
// Mirrors the reference's own interface for the BatchOptimizer path — same class
// and method names, argument order, enum values and error behaviour — so code
// (and tests) written against yangjames/Calico's C++ API read the same:
//   calico::BatchOptimizer              calico/batch_optimizer.h:23-73
//   calico::sensors::Sensor             calico/sensors/sensor_base.h:22-102
//   calico::sensors::Camera / Gyroscope / Accelerometer   calico/sensors/*.h
//   calico::Trajectory, WorldModel, RigidBody, Landmark, Pose3d
// What differs, because Eigen / abseil / Ceres are not dependencies here:
//   Eigen::VectorXd -> std::vector<double>, Eigen::Vector3d -> calico::Vector3d,
//   absl::Status(Or) -> calico::Status(Or) (absl code numbering),
//   ceres::Problem -> calico::Problem (records blocks by pointer like Ceres and
//   forwards them to the C ABI), ceres::Solver::Options/Summary -> the C structs.
// Link with -lcalico_hip. There is no CPU fallback: Optimize() fails with
// kInternal when no HIP device is usable.
#ifndef CALICO_CALICO_HPP_
#define CALICO_CALICO_HPP_

#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../calico_hip.h"
#include "../../calico_amd/csrc/device_math.hpp"  // sensor models shared with the device code

namespace calico {

// ---------------------------------------------------------------------------
// Status (absl numbering)
// ---------------------------------------------------------------------------
enum class StatusCode : int { kOk = 0, kInvalidArgument = 3, kFailedPrecondition = 9, kUnimplemented = 12, kInternal = 13 };
class Status {
 public:
  Status() = default;
  Status(StatusCode c, std::string m) : code_(c), msg_(std::move(m)) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const std::string& message() const { return msg_; }
 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(const std::string& m) { return Status(StatusCode::kInvalidArgument, m); }
inline Status FailedPreconditionError(const std::string& m) { return Status(StatusCode::kFailedPrecondition, m); }
inline Status InternalError(const std::string& m) { return Status(StatusCode::kInternal, m); }
template <class T> class StatusOr {
 public:
  StatusOr(const Status& s) : st_(s) {}  // NOLINT
  StatusOr(const T& v) : v_(v) {}        // NOLINT
  StatusOr(T&& v) : v_(std::move(v)) {}  // NOLINT
  bool ok() const { return st_.ok(); }
  const Status& status() const { return st_; }
  T& value() { return *v_; }
  const T& value() const { return *v_; }
  T& operator*() { return *v_; }
  const T& operator*() const { return *v_; }
  T* operator->() { return &*v_; }
  const T* operator->() const { return &*v_; }
 private:
  Status st_;
  std::optional<T> v_;
};

// ---------------------------------------------------------------------------
// Small fixed-size algebra (stand-ins for the Eigen types of typedefs.h)
// ---------------------------------------------------------------------------
using VectorXd = std::vector<double>;
struct Vector2d {
  double v[2] = {0, 0};
  Vector2d() = default;
  Vector2d(double a, double b) : v{a, b} {}
  double& x() { return v[0]; } double& y() { return v[1]; }
  double x() const { return v[0]; } double y() const { return v[1]; }
  double* data() { return v; } const double* data() const { return v; }
};
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double& x() { return v[0]; } double& y() { return v[1]; } double& z() { return v[2]; }
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
  double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; }
  double* data() { return v; } const double* data() const { return v; }
  int size() const { return 3; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};
inline Vector3d operator+(const Vector3d& a, const Vector3d& b) { return Vector3d(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vector3d operator-(const Vector3d& a, const Vector3d& b) { return Vector3d(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vector3d operator-(const Vector3d& a) { return Vector3d(-a[0], -a[1], -a[2]); }
inline Vector3d operator*(double s, const Vector3d& a) { return Vector3d(s * a[0], s * a[1], s * a[2]); }
inline Vector3d cross(const Vector3d& a, const Vector3d& b) {
  return Vector3d(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
// Eigen::Quaterniond: coeffs() storage x,y,z,w; constructor order w,x,y,z.
struct Quaterniond {
  double c[4] = {0, 0, 0, 1};
  Quaterniond() = default;
  Quaterniond(double w, double x, double y, double z) : c{x, y, z, w} {}
  double& x() { return c[0]; } double& y() { return c[1]; } double& z() { return c[2]; } double& w() { return c[3]; }
  double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
  struct Coeffs { double* p; double* data() { return p; } int size() const { return 4; } };
  Coeffs coeffs() { return Coeffs{c}; }
  const double* data() const { return c; }
  void setIdentity() { c[0] = c[1] = c[2] = 0; c[3] = 1; }
  Quaterniond conjugate() const { return Quaterniond(w(), -x(), -y(), -z()); }
  Quaterniond inverse() const {
    const double n2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    return n2 > 0 ? Quaterniond(w() / n2, -x() / n2, -y() / n2, -z() / n2) : Quaterniond(0, 0, 0, 0);
  }
  Quaterniond operator*(const Quaterniond& b) const {
    return Quaterniond(w() * b.w() - x() * b.x() - y() * b.y() - z() * b.z(), w() * b.x() + x() * b.w() + y() * b.z() - z() * b.y(),
                       w() * b.y() + y() * b.w() + z() * b.x() - x() * b.z(), w() * b.z() + z() * b.w() + x() * b.y() - y() * b.x());
  }
  Vector3d operator*(const Vector3d& v) const {  // Eigen _transformVector
    const Vector3d u(x(), y(), z());
    Vector3d uv = cross(u, v);
    uv = uv + uv;
    return v + w() * uv + cross(u, uv);
  }
  // AngleAxisd(angle, axis) -> quaternion
  static Quaterniond FromAngleAxis(double angle, const Vector3d& axis) {
    const double s = std::sin(0.5 * angle);
    return Quaterniond(std::cos(0.5 * angle), s * axis[0], s * axis[1], s * axis[2]);
  }
};

/// typedefs.h:38-153
class Pose3d {
 public:
  Pose3d() = default;
  Pose3d(const Quaterniond& q, const Vector3d& t) : q_(q), t_(t) {}
  Quaterniond& rotation() { return q_; }
  const Quaterniond& rotation() const { return q_; }
  Vector3d& translation() { return t_; }
  const Vector3d& translation() const { return t_; }
  /// [w, x, y, z], normalised (typedefs.h:69-75)
  void SetRotation(const std::array<double, 4>& q) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q_ = Quaterniond(q[0] / n, q[1] / n, q[2] / n, q[3] / n);
  }
  std::array<double, 4> GetRotation() const { return {q_.w(), q_.x(), q_.y(), q_.z()}; }
  void SetTranslation(const Vector3d& t) { t_ = t; }
  Vector3d GetTranslation() const { return t_; }
  Pose3d operator*(const Pose3d& T_b_a) const { return Pose3d(q_ * T_b_a.q_, q_ * T_b_a.t_ + t_); }
  Vector3d operator*(const Vector3d& p) const { return q_ * p + t_; }
  Pose3d inverse() const { const Quaterniond qi = q_.conjugate(); return Pose3d(qi, -(qi * t_)); }
 private:
  Quaterniond q_;
  Vector3d t_;
};

namespace utils {
/// optimization_utils.h:15-22
enum class LossFunctionType : int { kNone = 0, kHuber = 1, kCauchy = 2 };
}  // namespace utils

using SolverOptions = calico_solver_options;
struct Summary : calico_summary {
  std::string BriefReport() const {
    return "Calico-HIP Solver Report: Iterations: " + std::to_string(num_iterations) + ", Initial cost: " +
           std::to_string(initial_cost) + ", Final cost: " + std::to_string(final_cost) + ", Termination: " +
           (termination_type == CALICO_CONVERGENCE ? "CONVERGENCE" : (termination_type == CALICO_NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE"));
  }
  std::string FullReport() const { return BriefReport() + " (" + message + ")"; }
  bool IsSolutionUsable() const { return termination_type == CALICO_CONVERGENCE || termination_type == CALICO_NO_CONVERGENCE; }
};
/// batch_optimizer.cpp:10-17
inline SolverOptions DefaultSolverOptions() { SolverOptions o; calico_default_solver_options(&o); return o; }

// ---------------------------------------------------------------------------
// Problem: what ceres::Problem is to the reference. Blocks are identified by
// their address (Ceres semantics); Solve() flattens everything through the C
// ABI, runs the device LM and writes the estimates back IN PLACE.
// ---------------------------------------------------------------------------
class Problem {
 public:
  Problem() = default;
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;
  ~Problem() { if (h_) calico_problem_destroy(h_); }

  void AddParameterBlock(double* values, int size, int manifold = CALICO_MANIFOLD_EUCLIDEAN) {
    if (index_.count(values)) return;
    index_[values] = int(blocks_.size());
    blocks_.push_back({values, size, manifold, false});
  }
  void SetParameterBlockConstant(double* values) { blocks_[size_t(index_.at(values))].constant = true; }
  int NumParameterBlocks() const { return int(blocks_.size()); }
  int NumResidualBlocks() const { int n = 0; for (const auto& s : sensors_) n += int(s.stamps.size()); return n; }
  int NumResiduals() const { int n = 0; for (const auto& s : sensors_) n += int(s.stamps.size()) * (s.kind == CALICO_SENSOR_CAMERA ? 2 : 3); return n; }

  void SetSpline(int order, const std::vector<double>& knots, const std::vector<double>& basis, const std::vector<double*>& ctrl) {
    order_ = order; knots_ = knots; basis_ = basis; ctrl_ = ctrl;
  }
  int AddSensor(int kind, int model, double* intr, double* q, double* t, double* latency, double* gravity, double sigma,
                int loss, double loss_scale) {
    sensors_.push_back({kind, model, intr, q, t, latency, gravity, sigma, loss, loss_scale, {}, {}, {}, {}, {}});
    return int(sensors_.size()) - 1;
  }
  void AddCameraResidual(int sensor, const Vector2d& pixel, double stamp, double* point, double* body_q, double* body_t) {
    SensorRec& s = sensors_[size_t(sensor)];
    s.meas.push_back(pixel.x()); s.meas.push_back(pixel.y()); s.stamps.push_back(stamp);
    s.point.push_back(point); s.body_q.push_back(body_q); s.body_t.push_back(body_t);
  }
  void AddImuResidual(int sensor, const Vector3d& m, double stamp) {
    SensorRec& s = sensors_[size_t(sensor)];
    s.meas.push_back(m[0]); s.meas.push_back(m[1]); s.meas.push_back(m[2]); s.stamps.push_back(stamp);
  }

  /// ceres::Solve(options, &problem, &summary)
  Status Solve(const SolverOptions& options, Summary* summary, int device = 0) {
    if (h_) { calico_problem_destroy(h_); h_ = nullptr; }
    if (calico_problem_create(&h_, device) != CALICO_OK) return InternalError("calico_problem_create failed: no usable HIP device");
    ids_.assign(blocks_.size(), -1);
    for (size_t i = 0; i < blocks_.size(); ++i) {
      const BlockRec& b = blocks_[i];
      if (int st = calico_problem_add_param_block(h_, b.ptr, b.size, b.manifold, b.constant ? 1 : 0, &ids_[i])) return Err(st);
    }
    std::vector<int32_t> ctrl_ids;
    for (double* p : ctrl_) ctrl_ids.push_back(ids_[size_t(index_.at(p))]);
    if (int st = calico_problem_set_spline(h_, order_, int(knots_.size()), knots_.data(), basis_.data(), ctrl_ids.data())) return Err(st);
    std::map<std::pair<double*, double*>, int> bodies;
    sensor_ids_.clear();
    for (SensorRec& s : sensors_) {
      int sid = -1;
      if (int st = calico_problem_add_sensor(h_, s.kind, s.model, Id(s.intr), Id(s.q), Id(s.t), Id(s.latency),
                                             s.gravity ? Id(s.gravity) : -1, s.sigma, s.loss, s.loss_scale, &sid)) return Err(st);
      sensor_ids_.push_back(sid);
      const int64_t n = int64_t(s.stamps.size());
      if (s.kind == CALICO_SENSOR_CAMERA) {
        std::vector<int32_t> body(static_cast<size_t>(n)), point(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; ++i) {
          const auto key = std::make_pair(s.body_q[size_t(i)], s.body_t[size_t(i)]);
          auto it = bodies.find(key);
          if (it == bodies.end()) {
            int bid = -1;
            if (int st = calico_problem_add_rigid_body(h_, Id(key.first), Id(key.second), &bid)) return Err(st);
            it = bodies.emplace(key, bid).first;
          }
          body[size_t(i)] = it->second; point[size_t(i)] = Id(s.point[size_t(i)]);
        }
        if (int st = calico_problem_add_camera_residuals(h_, sid, n, s.meas.data(), s.stamps.data(), body.data(), point.data())) return Err(st);
      } else {
        if (int st = calico_problem_add_imu_residuals(h_, sid, n, s.meas.data(), s.stamps.data())) return Err(st);
      }
    }
    if (int st = calico_solve(h_, &options, summary)) return Err(st);
    for (size_t i = 0; i < blocks_.size(); ++i)  // in-place update, as Ceres does
      if (int st = calico_get_param_block(h_, ids_[i], blocks_[i].ptr)) return Err(st);
    return OkStatus();
  }
  /// problem.EvaluateResidualBlock(id, /*apply_loss_function=*/false, ...) for every block of a sensor.
  Status EvaluateResiduals(int sensor, std::vector<double>* out, std::vector<uint8_t>* valid) {
    if (!h_) return FailedPreconditionError("problem has not been solved");
    const SensorRec& s = sensors_[size_t(sensor)];
    const size_t n = s.stamps.size();
    out->assign(n * (s.kind == CALICO_SENSOR_CAMERA ? 2 : 3), 0.0); valid->assign(n, 0);
    const int st = calico_get_residuals(h_, sensor_ids_[size_t(sensor)], out->data(), valid->data());
    return st == CALICO_OK ? OkStatus() : Err(st);
  }
  calico_problem* handle() { return h_; }

 private:
  struct BlockRec { double* ptr; int size; int manifold; bool constant; };
  struct SensorRec {
    int kind, model; double *intr, *q, *t, *latency, *gravity; double sigma; int loss; double loss_scale;
    std::vector<double> meas, stamps; std::vector<double*> point, body_q, body_t;
  };
  int Id(double* p) const { return ids_[size_t(index_.at(p))]; }
  Status Err(int st) const { return Status(static_cast<StatusCode>(st), calico_last_error(h_)); }
  std::vector<BlockRec> blocks_;
  std::unordered_map<double*, int> index_;
  std::vector<int32_t> ids_;
  std::vector<SensorRec> sensors_;
  std::vector<int> sensor_ids_;
  int order_ = 0;
  std::vector<double> knots_, basis_;
  std::vector<double*> ctrl_;
  calico_problem* h_ = nullptr;
};

namespace utils {
/// optimization_utils.h:51-68
inline int AddPoseToProblem(Problem& problem, Pose3d& pose) {
  problem.AddParameterBlock(pose.translation().data(), 3);
  problem.AddParameterBlock(pose.rotation().coeffs().data(), 4, CALICO_MANIFOLD_EIGEN_QUATERNION);
  return 7;
}
inline void SetPoseConstantInProblem(Problem& problem, Pose3d& pose) {
  problem.SetParameterBlockConstant(pose.translation().data());
  problem.SetParameterBlockConstant(pose.rotation().coeffs().data());
}
}  // namespace utils

// ---------------------------------------------------------------------------
// Trajectory: 6-DOF B-spline of [axis-angle; position] (trajectory.{h,cpp}, bspline.{h,hpp})
// ---------------------------------------------------------------------------
class BSpline6 {
 public:
#ifdef CALICO_TEST_HOOKS
  /// Test builds only (-DCALICO_TEST_HOOKS, tests/cpp): the least-squares solve behind FitToData can be replaced, so
  /// that the host-only unit test runs the surrounding logic on a machine without a GPU with a checker-backed solver.
  /// A product build calls calico_fit_spline on device 0 -- the library has no host fallback.
  using FitSolver = int32_t (*)(int32_t order, int32_t n_knots, const double* knots, const double* basis, int64_t n,
                                const double* stamps, const double* data6, double* ctrl_out);
  static FitSolver& fit_solver() {
    static FitSolver solver = nullptr;
    return solver;
  }
#endif
  int GetSplineOrder() const { return order_; }
  const std::vector<double>& knots() const { return knots_; }
  const std::vector<double>& valid_knots() const { return valid_knots_; }
  std::vector<std::array<double, 6>>& control_points() { return ctrl_; }
  const std::vector<std::array<double, 6>>& control_points() const { return ctrl_; }
  const std::vector<double>& basis_matrices() const { return basis_; }  // n_seg × k × k
  /// bspline.hpp:138-150
  int GetSplineIndex(double t) const {
    if (t == valid_knots_.back()) return int(valid_knots_.size()) - 2;
    if (t < valid_knots_.back()) return int(std::upper_bound(valid_knots_.begin(), valid_knots_.end(), t) - valid_knots_.begin()) - 1;
    return -1;
  }
  int GetKnotIndexFromSplineIndex(int i) const { return i + order_ - 1; }
  int AddParametersToProblem(Problem& problem) {  // bspline.hpp:10-17
    int n = 0;
    for (auto& c : ctrl_) { problem.AddParameterBlock(c.data(), 6); n += 6; }
    return n;
  }
  /// bspline.hpp:19-37,163-297 (knot vector and basis matrices here, the least-squares solve on the device)
  Status FitToData(const std::vector<double>& time, const std::vector<std::array<double, 6>>& data, int order, double knot_frequency) {
    if (time.empty()) return InvalidArgumentError("Attempted to fit data on empty time vector.");
    if (data.empty()) return InvalidArgumentError("Attempted to fit on empty data.");
    if (time.size() != data.size()) return InvalidArgumentError("Data and time vectors are not the same size.");
    if (order < 2) return InvalidArgumentError("Spline order must be greater than 2. Got " + std::to_string(order));
    if (knot_frequency <= 0) return InvalidArgumentError("Knot frequency must be greater than 0.");
    order_ = order;
    const int deg = order - 1;
    const double duration = time.back() - time.front(), dt = 1.0 / knot_frequency;
    const int nvalid = 1 + int(std::ceil(duration * knot_frequency)), nk = nvalid + 2 * deg;
    knots_.assign(size_t(nk), 0.0); valid_knots_.assign(size_t(nvalid), 0.0);
    for (int i = -deg; i < nk - deg; ++i) {
      knots_[size_t(i + deg)] = time.front() + dt * i;
      if (i > -1 && i < nvalid) valid_knots_[size_t(i)] = knots_[size_t(i + deg)];
    }
    const int nseg = nvalid - 1;
    basis_.assign(size_t(nseg) * order * order, 0.0);
    for (int s = 0; s < nseg; ++s) {
      const std::vector<double> M = Basis(order, s + deg);
      std::copy(M.begin(), M.end(), basis_.begin() + size_t(s) * order * order);
    }
    // least-squares control points on the device (calico_fit_spline: banded normal equations + banded Cholesky; the
    // reference factors the dense design matrix by column-pivoted QR, bspline.hpp:287-293)
    const int ncp = nk - order, nd = int(time.size());
    std::vector<double> flat(size_t(nd) * 6), ctrl(size_t(ncp) * 6);
    for (int j = 0; j < nd; ++j) for (int c = 0; c < 6; ++c) flat[size_t(j) * 6 + c] = data[size_t(j)][size_t(c)];
#ifdef CALICO_TEST_HOOKS
    const int32_t rc = fit_solver() ? fit_solver()(order, nk, knots_.data(), basis_.data(), nd, time.data(), flat.data(), ctrl.data())
                                    : calico_fit_spline(0, order, nk, knots_.data(), basis_.data(), nd, time.data(), flat.data(), ctrl.data());
#else
    const int32_t rc = calico_fit_spline(0, order, nk, knots_.data(), basis_.data(), nd, time.data(), flat.data(), ctrl.data());
#endif
    if (rc == CALICO_INVALID_ARGUMENT) return InvalidArgumentError("spline fit: stamps must be sorted and inside the knot range");
    if (rc != CALICO_OK) return InternalError("spline fit failed on the device");
    ctrl_.assign(size_t(ncp), {});
    for (int i = 0; i < ncp; ++i) for (int c = 0; c < 6; ++c) ctrl_[size_t(i)][size_t(c)] = ctrl[size_t(i) * 6 + c];
    return OkStatus();
  }
  /// bspline.hpp:39-72 at one time; error codes of bspline.hpp:74-85
  Status Evaluate(double t, int derivative, double out[6]) const {
    if (derivative < 0 || derivative > order_ - 1) return InvalidArgumentError("Invalid derivative for interpolation.");
    if (t < valid_knots_.front() || t > valid_knots_.back())
      return InvalidArgumentError("Cannot interpolate " + std::to_string(t) + ". Value is not within valid knots.");
    const int si = GetSplineIndex(t), ki = si + order_ - 1, k = order_;
    const double dti = 1.0 / (knots_[size_t(ki + 1)] - knots_[size_t(ki)]), u = (t - knots_[size_t(ki)]) * dti;
    double scale = 1.0; for (int j = 0; j < derivative; ++j) scale *= dti;
    std::vector<double> U(size_t(k), 0.0);
    for (int i = derivative; i < k; ++i) { double coeff = 1.0; for (int j = i - derivative; j < i; ++j) coeff *= (j + 1); U[size_t(i)] = coeff * std::pow(u, i - derivative) * scale; }
    for (int c = 0; c < 6; ++c) out[c] = 0.0;
    for (int j = 0; j < k; ++j) {
      double w = 0.0; for (int i = 0; i < k; ++i) w += U[size_t(i)] * basis_[(size_t(si) * k + i) * k + j];
      for (int c = 0; c < 6; ++c) out[c] += w * ctrl_[size_t(si + j)][size_t(c)];
    }
    return OkStatus();
  }
 private:
  std::vector<double> Basis(int k, int i) const {  // bspline.hpp:191-244
    if (k == 1) return {1.0};
    const std::vector<double> P = Basis(k - 1, i);
    std::vector<double> M(size_t(k) * k, 0.0);
    for (int index = 0; index < k - 1; ++index) {
      const int j = i - k + 2 + index;
      const double den = knots_[size_t(j + k - 1)] - knots_[size_t(j)];
      const double d0 = den <= 0 ? 0.0 : (knots_[size_t(i)] - knots_[size_t(j)]) / den;
      const double d1 = den <= 0 ? 0.0 : (knots_[size_t(i + 1)] - knots_[size_t(i)]) / den;
      for (int r = 0; r < k; ++r) {
        const double m1 = r < k - 1 ? P[size_t(r) * (k - 1) + index] : 0.0;  // [M;0]
        const double m2 = r > 0 ? P[size_t(r - 1) * (k - 1) + index] : 0.0;  // [0;M]
        M[size_t(r) * k + index] += m1 * (1.0 - d0) - m2 * d1;
        M[size_t(r) * k + index + 1] += m1 * d0 + m2 * d1;
      }
    }
    return M;
  }
  int order_ = 0;
  std::vector<double> knots_, valid_knots_, basis_;
  std::vector<std::array<double, 6>> ctrl_;
};

class Trajectory {
 public:
  static constexpr int kDefaultSplineOrder = 6;       // trajectory.h:28
  static constexpr double kDefaultKnotFrequency = 10;  // trajectory.h:31
  /// trajectory.cpp:14-49
  Status FitSpline(const std::map<double, Pose3d>& poses_world_sensorrig, double knot_frequency = kDefaultKnotFrequency,
                   int spline_order = kDefaultSplineOrder) {
    poses_ = poses_world_sensorrig;
    std::vector<double> stamps;
    std::vector<std::array<double, 6>> data;
    std::array<double, 3> prev{};
    bool first = true;
    for (const auto& [stamp, T] : poses_) {  // std::map iterates sorted (trajectory.cpp:24 sorts)
      const Quaterniond& q = T.rotation();
      const double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
      std::array<double, 3> phi{0, 0, 0};
      if (n != 0.0) {  // Eigen::AngleAxisd(Quaterniond)
        const double angle = 2.0 * std::atan2(n, std::fabs(q.w())), s = (q.w() < 0 ? -1.0 : 1.0) / n;
        phi = {q.x() * s * angle, q.y() * s * angle, q.z() * s * angle};
      }
      if (!first) {  // UnwrapPhaseLogMap, trajectory.cpp:81-93
        const double theta = std::sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
        if (theta != 0) {
          const double k = std::round((phi[0] * prev[0] + phi[1] * prev[1] + phi[2] * prev[2] - theta * theta) / (2.0 * M_PI * theta));
          for (double& p : phi) p *= (1.0 + 2.0 * M_PI * k / theta);
        }
      }
      prev = phi; first = false;
      stamps.push_back(stamp);
      data.push_back({phi[0], phi[1], phi[2], T.translation()[0], T.translation()[1], T.translation()[2]});
    }
    return spline_.FitToData(stamps, data, spline_order, knot_frequency);
  }
  int AddParametersToProblem(Problem& problem) {  // trajectory.cpp:51 (+ the evaluation tables of :63-79)
    const int n = spline_.AddParametersToProblem(problem);
    std::vector<double*> ctrl;
    for (auto& c : spline_.control_points()) ctrl.push_back(c.data());
    problem.SetSpline(spline_.GetSplineOrder(), spline_.knots(), spline_.basis_matrices(), ctrl);
    return n;
  }
  const BSpline6& spline() const { return spline_; }
  BSpline6& spline() { return spline_; }
  /// trajectory.h:93-101
  static Pose3d VectorToPose3(const double v[6]) {
    const cal::Q4 q = cal::angle_axis_to_quat(cal::mk(v[0], v[1], v[2]));
    return Pose3d(Quaterniond(q.w, q.x, q.y, q.z), Vector3d(v[3], v[4], v[5]));
  }
  StatusOr<std::vector<Pose3d>> Interpolate(const std::vector<double>& interp_times) const {
    std::vector<Pose3d> out(interp_times.size());
    for (size_t i = 0; i < interp_times.size(); ++i) {
      double v[6];
      const Status st = spline_.Evaluate(interp_times[i], 0, v);
      if (!st.ok()) return st;
      out[i] = VectorToPose3(v);
    }
    return out;
  }
  const std::map<double, Pose3d>& trajectory() const { return poses_; }
 private:
  std::map<double, Pose3d> poses_;
  BSpline6 spline_;
};

// ---------------------------------------------------------------------------
// WorldModel (world_model.{h,cpp})
// ---------------------------------------------------------------------------
constexpr int kLandmarkFrameId = -1;
struct Landmark { Vector3d point; int id = 0; bool point_is_constant = false; };
struct RigidBody {
  std::unordered_map<int, Vector3d> model_definition;
  Pose3d T_world_rigidbody;
  int id = 0;
  bool world_pose_is_constant = false;
  bool model_definition_is_constant = false;
};
class WorldModel {
 public:
  static constexpr double kGravityDefaultZ = -9.80665;
  WorldModel() : gravity_(0.0, 0.0, kGravityDefaultZ) {}
  ~WorldModel() { Clear(); }
  Status AddLandmark(Landmark* landmark, bool take_ownership = true) {
    if (landmarks_.count(landmark->id)) return InvalidArgumentError("Landmark with id " + std::to_string(landmark->id) + " already exists in world model.");
    landmarks_[landmark->id] = landmark; own_l_[landmark->id] = take_ownership; return OkStatus();
  }
  Status AddRigidBody(RigidBody* rigidbody, bool take_ownership = true) {
    if (bodies_.count(rigidbody->id)) return InvalidArgumentError("Rigid body with id " + std::to_string(rigidbody->id) + "already exists in world model.");
    bodies_[rigidbody->id] = rigidbody; own_b_[rigidbody->id] = take_ownership; return OkStatus();
  }
  int AddParametersToProblem(Problem& problem) {  // world_model.cpp:40-77
    int n = 0;
    for (auto& [_, l] : landmarks_) {
      problem.AddParameterBlock(l->point.data(), 3); n += 3;
      if (l->point_is_constant) problem.SetParameterBlockConstant(l->point.data());
    }
    for (auto& [_, b] : bodies_) {
      for (auto& [__, p] : b->model_definition) { problem.AddParameterBlock(p.data(), 3); n += 3; }
      n += utils::AddPoseToProblem(problem, b->T_world_rigidbody);
      if (b->model_definition_is_constant) for (auto& [__, p] : b->model_definition) problem.SetParameterBlockConstant(p.data());
      if (b->world_pose_is_constant) utils::SetPoseConstantInProblem(problem, b->T_world_rigidbody);
    }
    problem.AddParameterBlock(gravity_.data(), 3); n += 3;
    if (!gravity_enabled_) problem.SetParameterBlockConstant(gravity_.data());
    return n;
  }
  std::map<int, Landmark*>& landmarks() { return landmarks_; }
  const std::map<int, Landmark*>& landmarks() const { return landmarks_; }
  std::map<int, RigidBody*>& rigidbodies() { return bodies_; }
  const std::map<int, RigidBody*>& rigidbodies() const { return bodies_; }
  void EnableGravityEstimation(bool) { /* a no-op in the reference too (world_model.cpp:79-81, quirk Q6) */ }
  Vector3d& gravity() { return gravity_; }
  const Vector3d& gravity() const { return gravity_; }
  void SetGravity(const Vector3d& g) { gravity_ = g; }
  const Vector3d& GetGravity() const { return gravity_; }
  int NumberOfLandmarks() const { return int(landmarks_.size()); }
  int NumberOfRigidBodies() const { return int(bodies_.size()); }
  void ClearLandmarks() { for (auto& [id, l] : landmarks_) if (own_l_[id]) delete l; landmarks_.clear(); own_l_.clear(); }
  void ClearRigidBodies() { for (auto& [id, b] : bodies_) if (own_b_[id]) delete b; bodies_.clear(); own_b_.clear(); }
  void Clear() { ClearLandmarks(); ClearRigidBodies(); }
 private:
  std::map<int, Landmark*> landmarks_;
  std::map<int, RigidBody*> bodies_;
  std::map<int, bool> own_l_, own_b_;
  Vector3d gravity_;
  bool gravity_enabled_ = false;
};

// ---------------------------------------------------------------------------
// Sensors (sensor_base.h, camera.{h,cpp}, gyroscope.{h,cpp}, accelerometer.{h,cpp})
// ---------------------------------------------------------------------------
namespace sensors {

enum class CameraIntrinsicsModel : int { kNone = 0, kOpenCv5, kOpenCv8, kKannalaBrandt, kDoubleSphere, kFieldOfView, kUnifiedCamera, kExtendedUnifiedCamera };
enum class GyroscopeIntrinsicsModel : int { kNone = 0, kGyroscopeScaleOnly, kGyroscopeScaleAndBias, kGyroscopeVectorNav };
enum class AccelerometerIntrinsicsModel : int { kNone = 0, kAccelerometerScaleOnly, kAccelerometerScaleAndBias, kAccelerometerVectorNav };
inline int NumberOfParameters(CameraIntrinsicsModel m) { static const int k[] = {-1, 8, 11, 7, 5, 4, 4, 5}; return k[int(m)]; }
inline int NumberOfImuParameters(int m) { return m == 1 ? 1 : (m == 2 ? 4 : (m == 3 ? 12 : -1)); }

/// sensor_base.h:22-102 — the plugin contract consumed by BatchOptimizer.
class Sensor {
 public:
  virtual ~Sensor() = default;
  virtual void SetName(const std::string& name) = 0;
  virtual const std::string& GetName() const = 0;
  virtual void SetExtrinsics(const Pose3d& T_sensorrig_sensor) = 0;
  virtual const Pose3d& GetExtrinsics() const = 0;
  virtual Status SetIntrinsics(const VectorXd& intrinsics) = 0;
  virtual const VectorXd& GetIntrinsics() const = 0;
  virtual Status SetLatency(double latency) = 0;
  virtual double GetLatency() const = 0;
  virtual void EnableExtrinsicsEstimation(bool enable) = 0;
  virtual void EnableIntrinsicsEstimation(bool enable) = 0;
  virtual void EnableLatencyEstimation(bool enable) = 0;
  virtual Status UpdateResiduals(Problem& problem) = 0;
  virtual void ClearResidualInfo() = 0;
  virtual void SetLossFunction(utils::LossFunctionType loss, double scale = 1.0) = 0;
  virtual StatusOr<int> AddParametersToProblem(Problem& problem) = 0;
  virtual StatusOr<int> AddResidualsToProblem(Problem& problem, Trajectory& sensorrig_trajectory, WorldModel& world_model) = 0;
  virtual Status SetMeasurementNoise(double sigma) = 0;
};

// Shared state/behaviour of the three sensors. The reference leaves the three enable flags and the
// loss type uninitialised (quirk Q1); here they default to false / kNone.
class SensorCommon : public Sensor {
 public:
  void SetName(const std::string& name) final { name_ = name; }
  const std::string& GetName() const final { return name_; }
  void SetExtrinsics(const Pose3d& T) final { T_sensorrig_sensor_ = T; }
  const Pose3d& GetExtrinsics() const final { return T_sensorrig_sensor_; }
  const VectorXd& GetIntrinsics() const final { return intrinsics_; }
  Status SetLatency(double latency) final { latency_ = latency; return OkStatus(); }
  double GetLatency() const final { return latency_; }
  void EnableExtrinsicsEstimation(bool e) final { extrinsics_enabled_ = e; }
  void EnableIntrinsicsEstimation(bool e) final { intrinsics_enabled_ = e; }
  void EnableLatencyEstimation(bool e) final { latency_enabled_ = e; }
  void SetLossFunction(utils::LossFunctionType loss, double scale = 1.0) final { loss_function_ = loss; loss_scale_ = scale; }
  Status SetMeasurementNoise(double sigma) final {  // camera.cpp:62-68
    if (sigma <= 0.0) return InvalidArgumentError("Sigma must be greater than 0.");
    sigma_ = sigma; return OkStatus();
  }
 protected:
  // camera.cpp:92-113 / gyroscope.cpp:10-31 / accelerometer.cpp:10-33
  StatusOr<int> AddCommonParameters(Problem& problem, bool model_set, const char* what) {
    if (!model_set) return FailedPreconditionError(std::string("Cannot add ") + what + " parameters. Model is not yet defined.");
    int n = 0;
    problem.AddParameterBlock(intrinsics_.data(), int(intrinsics_.size())); n += int(intrinsics_.size());
    n += utils::AddPoseToProblem(problem, T_sensorrig_sensor_);
    problem.AddParameterBlock(&latency_, 1); ++n;
    if (!intrinsics_enabled_) problem.SetParameterBlockConstant(intrinsics_.data());
    if (!extrinsics_enabled_) utils::SetPoseConstantInProblem(problem, T_sensorrig_sensor_);
    if (!latency_enabled_) problem.SetParameterBlockConstant(&latency_);
    return n;
  }
  std::string name_;
  bool intrinsics_enabled_ = false, extrinsics_enabled_ = false, latency_enabled_ = false;
  Pose3d T_sensorrig_sensor_;
  VectorXd intrinsics_;
  double latency_ = 0.0, sigma_ = 1.0;
  utils::LossFunctionType loss_function_ = utils::LossFunctionType::kNone;
  double loss_scale_ = 1.0;
  int problem_sensor_ = -1;
};

/// camera.h:24-58
struct CameraObservationId {
  double stamp; int image_id; int model_id; int feature_id;
  bool operator==(const CameraObservationId& o) const { return stamp == o.stamp && image_id == o.image_id && model_id == o.model_id && feature_id == o.feature_id; }
};
struct CameraObservationIdHash {
  size_t operator()(const CameraObservationId& id) const {
    size_t h = std::hash<double>()(id.stamp);
    for (int v : {id.image_id, id.model_id, id.feature_id}) h = h * 1000003u ^ std::hash<int>()(v);
    return h;
  }
};
struct CameraMeasurement { Vector2d pixel; CameraObservationId id; };

inline bool ProjectPointHost(CameraIntrinsicsModel model, const double* k, const cal::V3& p, double pix[2]) {
  double D[2][3], dK[2][cal::kMaxIntr];
  switch (int(model)) {
    case 1: return cal::project<1, false>(k, p, pix, D, dK);
    case 2: return cal::project<2, false>(k, p, pix, D, dK);
    case 3: return cal::project<3, false>(k, p, pix, D, dK);
    case 4: return cal::project<4, false>(k, p, pix, D, dK);
    case 5: return cal::project<5, false>(k, p, pix, D, dK);
    case 6: return cal::project<6, false>(k, p, pix, D, dK);
    case 7: return cal::project<7, false>(k, p, pix, D, dK);
    default: return false;
  }
}

class Camera : public SensorCommon {
 public:
  Camera() = default;
  Camera(const Camera&) = delete;
  Camera& operator=(const Camera&) = delete;
  Status SetIntrinsics(const VectorXd& intrinsics) final {  // camera.cpp:24-37
    if (model_ == CameraIntrinsicsModel::kNone) return InvalidArgumentError("Camera model has not been set!");
    if (int(intrinsics.size()) != NumberOfParameters(model_))
      return InvalidArgumentError("Tried to set intrinsics of size " + std::to_string(intrinsics.size()) + " for camera " + GetName() +
                                  ". Expected intrinsics size of " + std::to_string(NumberOfParameters(model_)));
    intrinsics_ = intrinsics; return OkStatus();
  }
  Status SetModel(CameraIntrinsicsModel m) {  // camera.cpp:210-219
    if (int(m) < 1 || int(m) > 7) return InvalidArgumentError("Could not create camera model for type " + std::to_string(int(m)) + ". It is likely not yet implemented.");
    model_ = m; intrinsics_.assign(size_t(NumberOfParameters(m)), 0.0); return OkStatus();
  }
  CameraIntrinsicsModel GetModel() const { return model_; }
  StatusOr<int> AddParametersToProblem(Problem& problem) final {
    return AddCommonParameters(problem, model_ != CameraIntrinsicsModel::kNone, "camera");
  }
  /// camera.cpp:115-153
  StatusOr<int> AddResidualsToProblem(Problem& problem, Trajectory& sensorrig_trajectory, WorldModel& world_model) final {
    (void)sensorrig_trajectory;
    problem_sensor_ = problem.AddSensor(CALICO_SENSOR_CAMERA, int(model_), intrinsics_.data(), T_sensorrig_sensor_.rotation().coeffs().data(),
                                        T_sensorrig_sensor_.translation().data(), &latency_, nullptr, sigma_, int(loss_function_), loss_scale_);
    int added = 0;
    residual_order_.clear();
    for (const auto& [observation_id, measurement] : id_to_measurement_) {
      if (outlier_ids_.count(observation_id)) continue;
      const int rigidbody_id = observation_id.model_id;
      if (!world_model.rigidbodies().count(rigidbody_id))
        return FailedPreconditionError("Attempted to create cost function from an observation for a rigidbody with id " +
                                       std::to_string(rigidbody_id) + " that does not exist in the world model.");
      RigidBody* body = world_model.rigidbodies().at(rigidbody_id);
      Vector3d& t_model_point = body->model_definition.at(observation_id.feature_id);  // throws like the reference (camera.cpp:135)
      problem.AddCameraResidual(problem_sensor_, measurement.pixel, observation_id.stamp, t_model_point.data(),
                                body->T_world_rigidbody.rotation().coeffs().data(), body->T_world_rigidbody.translation().data());
      residual_order_.push_back(observation_id);
      ++added;
    }
    return added;
  }
  Status UpdateResiduals(Problem& problem) final {  // camera.cpp:70-80
    std::vector<double> r; std::vector<uint8_t> valid;
    const Status st = problem.EvaluateResiduals(problem_sensor_, &r, &valid);
    if (!st.ok()) return InternalError("Failed to update residual for camera " + name_);
    for (size_t i = 0; i < residual_order_.size(); ++i) id_to_residual_[residual_order_[i]] = Vector2d(r[2 * i], r[2 * i + 1]);
    return OkStatus();
  }
  void ClearResidualInfo() final { residual_order_.clear(); id_to_residual_.clear(); }
  /// camera.cpp:155-208
  StatusOr<std::vector<CameraMeasurement>> Project(const std::vector<double>& interp_times, const Trajectory& sensorrig_trajectory,
                                                   const WorldModel& world_model) const {
    auto poses = sensorrig_trajectory.Interpolate(interp_times);
    if (!poses.ok()) return poses.status();
    std::vector<CameraMeasurement> out;
    int image_id = 0;
    for (size_t i = 0; i < interp_times.size(); ++i) {
      const Pose3d T_camera_world = ((*poses)[i] * T_sensorrig_sensor_).inverse();
      auto emit = [&](const Vector3d& pc, int model_id, int feature_id) {
        if (pc.z() <= 0) return;
        double pix[2] = {0, 0};
        ProjectPointHost(model_, intrinsics_.data(), cal::mk(pc[0], pc[1], pc[2]), pix);
        out.push_back({Vector2d(pix[0], pix[1]), {interp_times[i] + latency_, image_id, model_id, feature_id}});
      };
      for (const auto& [id, l] : world_model.landmarks()) emit(T_camera_world * l->point, kLandmarkFrameId, id);
      for (const auto& [bid, body] : world_model.rigidbodies()) {
        const Pose3d T_camera_body = T_camera_world * body->T_world_rigidbody;
        std::vector<int> keys;
        for (const auto& kv : body->model_definition) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());
        for (int pid : keys) emit(T_camera_body * body->model_definition.at(pid), bid, pid);
      }
      ++image_id;
    }
    return out;
  }
  Status AddMeasurement(const CameraMeasurement& m) {  // camera.cpp:226-236
    if (id_to_measurement_.count(m.id))
      return InvalidArgumentError("Tried to add redundant measurement - Image id: " + std::to_string(m.id.image_id) + ", model id: " +
                                  std::to_string(m.id.model_id) + ", feature id: " + std::to_string(m.id.feature_id));
    id_to_measurement_[m.id] = m; return OkStatus();
  }
  Status AddMeasurements(const std::vector<CameraMeasurement>& ms) {  // camera.cpp:238-251 (Q11: unique ones are still added)
    std::string message;
    for (const auto& m : ms) { const Status st = AddMeasurement(m); if (!st.ok()) message += st.message() + "\n"; }
    return message.empty() ? OkStatus() : InvalidArgumentError(message);
  }
  const std::unordered_map<CameraObservationId, CameraMeasurement, CameraObservationIdHash>& GetMeasurementIdToMeasurement() const { return id_to_measurement_; }
  StatusOr<std::vector<std::pair<CameraMeasurement, Vector2d>>> GetMeasurementResidualPairs() const {  // camera.cpp:258-279
    if (id_to_residual_.size() > id_to_measurement_.size()) return InternalError("There are more residuals than measurements.");
    if (id_to_measurement_.empty()) return FailedPreconditionError("Measurements are empty. Nothing to return.");
    std::vector<std::pair<CameraMeasurement, Vector2d>> pairs;
    for (const auto& [id, r] : id_to_residual_) {
      auto it = id_to_measurement_.find(id);
      if (it == id_to_measurement_.end()) return InternalError("Found a residual that doesn't correspond to any measurement.");
      pairs.push_back({it->second, r});
    }
    return pairs;
  }
  Status MarkOutlierById(const CameraObservationId& id) {  // camera.cpp:281-290
    if (!id_to_measurement_.count(id)) return InvalidArgumentError("Attempted to add id that is not within the measurement set.");
    outlier_ids_.insert(id); return OkStatus();
  }
  Status MarkOutliersById(const std::vector<CameraObservationId>& ids) { for (const auto& id : ids) { const Status st = MarkOutlierById(id); if (!st.ok()) return st; } return OkStatus(); }
  void ClearOutliersList() { outlier_ids_.clear(); }
  void ClearMeasurements() { id_to_measurement_.clear(); residual_order_.clear(); id_to_residual_.clear(); outlier_ids_.clear(); }
  int NumberOfMeasurements() const { return int(id_to_measurement_.size()); }
 private:
  CameraIntrinsicsModel model_ = CameraIntrinsicsModel::kNone;
  std::unordered_map<CameraObservationId, CameraMeasurement, CameraObservationIdHash> id_to_measurement_;
  std::unordered_map<CameraObservationId, Vector2d, CameraObservationIdHash> id_to_residual_;
  std::vector<CameraObservationId> residual_order_;
  std::unordered_set<CameraObservationId, CameraObservationIdHash> outlier_ids_;
};

/// gyroscope.h:20-44 / accelerometer.h (same shape)
struct ImuObservationId {
  double stamp; int sequence;
  bool operator==(const ImuObservationId& o) const { return stamp == o.stamp && sequence == o.sequence; }
};
struct ImuObservationIdHash { size_t operator()(const ImuObservationId& id) const { return std::hash<double>()(id.stamp) * 1000003u ^ std::hash<int>()(id.sequence); } };
struct ImuMeasurement { Vector3d measurement; ImuObservationId id; };
using GyroscopeObservationId = ImuObservationId;
using GyroscopeMeasurement = ImuMeasurement;
using AccelerometerObservationId = ImuObservationId;
using AccelerometerMeasurement = ImuMeasurement;

template <int KIND>
class ImuSensor : public SensorCommon {
 public:
  Status SetIntrinsics(const VectorXd& intrinsics) final {
    if (model_ == 0) return InvalidArgumentError("Model has not been set!");
    if (int(intrinsics.size()) != NumberOfImuParameters(model_))
      return InvalidArgumentError("Tried to set intrinsics of size " + std::to_string(intrinsics.size()) + " for sensor " + GetName() +
                                  ". Expected intrinsics size of " + std::to_string(NumberOfImuParameters(model_)));
    intrinsics_ = intrinsics; return OkStatus();
  }
  StatusOr<int> AddParametersToProblem(Problem& problem) final { return AddCommonParameters(problem, model_ != 0, KIND == CALICO_SENSOR_GYROSCOPE ? "gyroscope" : "accelerometer"); }
  /// gyroscope.cpp:33-54 / accelerometer.cpp:35-56
  StatusOr<int> AddResidualsToProblem(Problem& problem, Trajectory&, WorldModel& world_model) final {
    problem_sensor_ = problem.AddSensor(KIND, model_, intrinsics_.data(), T_sensorrig_sensor_.rotation().coeffs().data(),
                                        T_sensorrig_sensor_.translation().data(), &latency_,
                                        KIND == CALICO_SENSOR_ACCELEROMETER ? world_model.gravity().data() : nullptr, sigma_,
                                        int(loss_function_), loss_scale_);
    residual_order_.clear();
    for (const auto& [id, m] : id_to_measurement_) { problem.AddImuResidual(problem_sensor_, m.measurement, id.stamp); residual_order_.push_back(id); }
    return int(residual_order_.size());
  }
  Status UpdateResiduals(Problem& problem) final {
    std::vector<double> r; std::vector<uint8_t> valid;
    const Status st = problem.EvaluateResiduals(problem_sensor_, &r, &valid);
    if (!st.ok()) return InternalError("Failed to update residual for sensor " + name_);
    for (size_t i = 0; i < residual_order_.size(); ++i) id_to_residual_[residual_order_[i]] = Vector3d(r[3 * i], r[3 * i + 1], r[3 * i + 2]);
    return OkStatus();
  }
  void ClearResidualInfo() final { residual_order_.clear(); id_to_residual_.clear(); }
  Status AddMeasurement(const ImuMeasurement& m) {
    if (id_to_measurement_.count(m.id))
      return InvalidArgumentError("Tried to add redundant measurement - Sequence: " + std::to_string(m.id.sequence) + ", stamp: " + std::to_string(m.id.stamp));
    id_to_measurement_[m.id] = m; return OkStatus();
  }
  Status AddMeasurements(const std::vector<ImuMeasurement>& ms) {
    std::string message;
    for (const auto& m : ms) { const Status st = AddMeasurement(m); if (!st.ok()) message += st.message() + "\n"; }
    return message.empty() ? OkStatus() : InvalidArgumentError(message);
  }
  void ClearMeasurements() { id_to_measurement_.clear(); }
  int NumberOfMeasurements() const { return int(id_to_measurement_.size()); }
  const std::unordered_map<ImuObservationId, Vector3d, ImuObservationIdHash>& residuals() const { return id_to_residual_; }
  /// gyroscope.cpp:56-82 / accelerometer.cpp:76-123
  StatusOr<std::vector<ImuMeasurement>> Project(const std::vector<double>& interp_times, const Trajectory& traj, const WorldModel& world_model) const {
    std::vector<ImuMeasurement> out(interp_times.size());
    const Quaterniond& q = T_sensorrig_sensor_.rotation();
    cal::Q4 qs; qs.x = q.x(); qs.y = q.y(); qs.z = q.z(); qs.w = q.w();
    const cal::M3 R_rs = cal::rotmat(cal::normalized(qs));
    for (size_t i = 0; i < interp_times.size(); ++i) {
      double p[6], pd[6], pdd[6];
      Status st = traj.spline().Evaluate(interp_times[i], 0, p); if (!st.ok()) return st;
      st = traj.spline().Evaluate(interp_times[i], 1, pd); if (!st.ok()) return st;
      const cal::V3 phi = cal::mk(-p[0], -p[1], -p[2]), phid = cal::mk(-pd[0], -pd[1], -pd[2]);
      const cal::Rodrigues<double> R = cal::rodrigues<double>(phi.x, phi.y, phi.z, true);
      cal::V3 omega; cal::rod_J_apply<double>(R, phid.x, phid.y, phid.z, &omega.x, &omega.y, &omega.z);
      cal::V3 in;
      if (KIND == CALICO_SENSOR_GYROSCOPE) {
        in = -cal::mulT(R_rs, omega);
      } else {
        st = traj.spline().Evaluate(interp_times[i], 2, pdd); if (!st.ok()) return st;
        const cal::V3 phidd = cal::mk(-pdd[0], -pdd[1], -pdd[2]);
        double jdd[3], Hv[3][3];
        cal::rod_J_apply<double>(R, phidd.x, phidd.y, phidd.z, &jdd[0], &jdd[1], &jdd[2]);
        cal::rod_H_apply<double>(R, phid.x, phid.y, phid.z, Hv);
        const cal::V3 alpha = cal::mk(phid.x * Hv[0][0] + phid.y * Hv[1][0] + phid.z * Hv[2][0] + jdd[0],
                                      phid.x * Hv[0][1] + phid.y * Hv[1][1] + phid.z * Hv[2][1] + jdd[1],
                                      phid.x * Hv[0][2] + phid.y * Hv[1][2] + phid.z * Hv[2][2] + jdd[2]);
        const cal::M3 R_rw = cal::rotmat(cal::angle_axis_to_quat(phi));
        const Vector3d& g = world_model.gravity();
        const Vector3d& tt = T_sensorrig_sensor_.translation();
        const cal::V3 t = cal::mk(tt[0], tt[1], tt[2]);
        const cal::V3 b = cal::mul(R_rw, cal::mk(pdd[3] - g[0], pdd[4] - g[1], pdd[5] - g[2])) + cal::cross(omega, cal::cross(omega, t)) - cal::cross(alpha, t);
        in = cal::mulT(R_rs, b);
      }
      double f[3], Mw[3][3], dK[3][cal::kMaxIntr];
      cal::imu_project<false>(model_, intrinsics_.data(), in, f, Mw, dK);
      out[i] = {Vector3d(f[0], f[1], f[2]), {interp_times[i] + latency_, int(i)}};
    }
    return out;
  }
 protected:
  Status SetModelInt(int m) {
    if (m < 1 || m > 3) return InvalidArgumentError("Could not create model for type " + std::to_string(m) + ". It is likely not yet implemented.");
    model_ = m; intrinsics_.assign(size_t(NumberOfImuParameters(m)), 0.0); return OkStatus();
  }
  int model_ = 0;
  std::unordered_map<ImuObservationId, ImuMeasurement, ImuObservationIdHash> id_to_measurement_;
  std::unordered_map<ImuObservationId, Vector3d, ImuObservationIdHash> id_to_residual_;
  std::vector<ImuObservationId> residual_order_;
};

class Gyroscope : public ImuSensor<CALICO_SENSOR_GYROSCOPE> {
 public:
  Status SetModel(GyroscopeIntrinsicsModel m) { return SetModelInt(int(m)); }
  GyroscopeIntrinsicsModel GetModel() const { return static_cast<GyroscopeIntrinsicsModel>(model_); }
};
class Accelerometer : public ImuSensor<CALICO_SENSOR_ACCELEROMETER> {
 public:
  Status SetModel(AccelerometerIntrinsicsModel m) { return SetModelInt(int(m)); }
  AccelerometerIntrinsicsModel GetModel() const { return static_cast<AccelerometerIntrinsicsModel>(model_); }
};

}  // namespace sensors

// ---------------------------------------------------------------------------
// BatchOptimizer (batch_optimizer.{h,cpp})
// ---------------------------------------------------------------------------
class BatchOptimizer {
 public:
  ~BatchOptimizer() {  // batch_optimizer.cpp:19-33: un-owned pointers are released, owned ones deleted
    for (size_t i = 0; i < sensors_.size(); ++i) if (own_sensors_[i]) delete sensors_[i];
    if (own_world_model_) delete world_model_;
    if (own_trajectory_) delete trajectory_;
  }
  void AddSensor(sensors::Sensor* sensor, bool take_ownership = true) { sensors_.push_back(sensor); own_sensors_.push_back(take_ownership); }
  void AddWorldModel(WorldModel* world_model, bool take_ownership = true) { world_model_ = world_model; own_world_model_ = take_ownership; }
  void AddTrajectory(Trajectory* trajectory_world_sensorrig, bool take_ownership = true) { trajectory_ = trajectory_world_sensorrig; own_trajectory_ = take_ownership; }
  /// batch_optimizer.cpp:53-81
  StatusOr<Summary> Optimize(const SolverOptions& options = DefaultSolverOptions(), int device = 0) {
    if (!world_model_ || !trajectory_) return FailedPreconditionError("world model and trajectory must be added before Optimize()");
    Problem problem;
    world_model_->AddParametersToProblem(problem);
    trajectory_->AddParametersToProblem(problem);
    for (sensors::Sensor* sensor : sensors_) {
      sensor->ClearResidualInfo();
      const auto np = sensor->AddParametersToProblem(problem);
      if (!np.ok()) return np.status();
      const auto nr = sensor->AddResidualsToProblem(problem, *trajectory_, *world_model_);
      if (!nr.ok()) return nr.status();
    }
    Summary summary;
    const Status st = problem.Solve(options, &summary, device);
    if (!st.ok()) return st;
    for (sensors::Sensor* sensor : sensors_) {
      const Status us = sensor->UpdateResiduals(problem);
      if (!us.ok()) return us;
    }
    return summary;
  }
 private:
  std::vector<sensors::Sensor*> sensors_;
  std::vector<bool> own_sensors_;
  WorldModel* world_model_ = nullptr;
  Trajectory* trajectory_ = nullptr;
  bool own_world_model_ = false, own_trajectory_ = false;
};

}  // namespace calico

#endif  // CALICO_CALICO_HPP_
