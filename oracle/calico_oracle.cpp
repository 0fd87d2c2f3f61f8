// calico_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference hot path BatchOptimizer::Optimize
// (calico/batch_optimizer.cpp:53-81): flattened problem, Ceres-style
// residual/Jacobian evaluation by 4-wide forward-mode passes, loss
// correction, manifold projection, and the trust-region Levenberg–Marquardt
// loop with a dense Cholesky normal-equation solve. Also the synthetic
// measurement generators Sensor::Project (camera.cpp:155-208,
// gyroscope.cpp:56-82, accelerometer.cpp:76-123) and the spline fit, used to
// build test problems.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, load or call this library; libcalico_hip.so never links it.
//
// Ceres Solver is not vendored by the reference (CMakeLists.txt:15,
// unpinned, >= 2.1 implied by ceres::Manifold). Everything marked [Ceres]
// restates the published algorithm of Ceres 2.1/2.2 (trust_region_minimizer.cc,
// levenberg_marquardt_strategy.cc, corrector.cc, loss_function.cc,
// manifold.h, dynamic_autodiff_cost_function.h). PARITY UNPINNED for those
// parts: the reference's tests pin only the converged fixed point
// (batch_optimizer_test.cpp:185-210) and "cost == 0 at truth"
// (gyroscope_test.cpp:174-182, accelerometer_test.cpp:194-202).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../include/calico_hip.h"
#include "oracle_math.hpp"
#include "oracle_spline.hpp"
#include "../calico_amd/csrc/shard.hpp"  // the product's partition rule, exercised here on CPU

namespace oracle {

// Trust-region control of [Ceres] TrustRegionMinimizer / LevenbergMarquardtStrategy: the radius after an accepted step
// (StepAccepted: radius / max(1/3, 1 - (2 rho - 1)^3), capped), after a rejected one (StepRejected: radius /
// decrease_factor, the factor doubling with every consecutive rejection and going back to 2 on acceptance), after an
// invalid linear solve (StepIsInvalid: radius / 2, factor untouched), and the step quality rho -- the lowest double
// when the candidate's cost could not be evaluated. Pinned by the iteration table the reference ships
// (demos/imu_camera_calibration.ipynb; tests/test_ceres_log.py replays it through oracle_lm_control_replay).
struct TrustRegionControl {
  double radius, decrease_factor;
  static double relative_decrease(double x_cost, double candidate_cost, double model_cost_change) {
    return candidate_cost >= std::numeric_limits<double>::max() ? std::numeric_limits<double>::lowest()
                                                                 : (x_cost - candidate_cost) / model_cost_change;
  }
  void step_accepted(double rho, double max_radius) {
    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
    radius = std::min(max_radius, radius);
    decrease_factor = 2.0;
  }
  void step_rejected() { radius = radius / decrease_factor; decrease_factor *= 2.0; }
  void step_is_invalid() { radius *= 0.5; }
};

struct ParamBlock {
  std::vector<double> v;
  int size = 0;
  int manifold = 0;
  bool constant = false;
  bool used = false;
  int tangent_size() const { return manifold == CALICO_MANIFOLD_EIGEN_QUATERNION ? 3 : size; }
  int tangent_offset = -1;  // in the reduced problem; -1 if constant or unused
};

struct Body { int q, t; };

struct Sensor {
  int kind, model;
  int intr, q, t, lat, grav;
  double sigma, information;
  int loss; double loss_scale;
  std::vector<double> meas;    // n×dim
  std::vector<double> stamps;  // n
  std::vector<int> body, point, seg;
  int dim() const { return kind == CALICO_SENSOR_CAMERA ? 2 : 3; }
  int64_t n() const { return int64_t(stamps.size()); }
};

struct BlockRef { int sensor; int64_t obs; };

struct Problem {
  std::vector<ParamBlock> blocks;
  std::vector<Body> bodies;
  std::vector<Sensor> sensors;
  int order = 0;
  std::vector<double> knots, valid_knots, basis;
  std::vector<int> ctrl;
  std::string error;
  // reduced problem
  int n_eff = 0;
  std::vector<int> reduced_blocks;  // in tangent order
  std::vector<BlockRef> rblocks;
  std::vector<int64_t> res_offset;  // per residual block: first residual row
  int64_t n_res = 0;
  std::vector<calico_iteration> iterations;
  // multi-rank emulation (tests): this rank evaluates only its time window and sums with the others
  int rank = 0, world = 1;
  int32_t (*allreduce)(void*, double*, int64_t) = nullptr;
  void* allreduce_ctx = nullptr;
  std::vector<char> own;  // per residual block
  // wall time of the last Solve by part (bench.py's cpu_baseline): residual / Jacobian evaluation, normal-equation
  // assembly (Ceres: both inside the evaluator / the Schur eliminator's block products), dense factorisation + solve
  mutable double t_evaluate = 0, t_assemble = 0, t_linear_solve = 0;
  void reduce(double* buf, int64_t n) const { if (allreduce) allreduce(allreduce_ctx, buf, n); }
  int set_error(int code, const std::string& m) { error = m; return code; }
};

// bspline.hpp:138-150 on the valid knots.
static int GetSplineIndex(const Problem& P, double t) {
  const std::vector<double>& vk = P.valid_knots;
  if (t == vk.back()) return int(vk.size()) - 2;
  if (t < vk.back()) {
    auto it = std::upper_bound(vk.begin(), vk.end(), t);
    return int(it - vk.begin()) - 1;
  }
  return -1;
}

// Parameter-block list of one residual block, in the order of the reference's
// CreateCostFunction (camera_cost_functor.cpp:28-60, gyroscope_cost_functor.cpp:27-47,
// accelerometer_cost_functor.cpp:28-51).
static int BlockParams(const Problem& P, const Sensor& s, int64_t i, int ids[32]) {
  int n = 0;
  ids[n++] = s.intr; ids[n++] = s.q; ids[n++] = s.t; ids[n++] = s.lat;
  if (s.kind == CALICO_SENSOR_CAMERA) {
    ids[n++] = s.point[i]; ids[n++] = P.bodies[s.body[i]].q; ids[n++] = P.bodies[s.body[i]].t;
  } else if (s.kind == CALICO_SENSOR_ACCELEROMETER) {
    ids[n++] = s.grav;
  }
  for (int j = 0; j < P.order; ++j) ids[n++] = P.ctrl[s.seg[i] + j];
  return n;
}

static EvalParams MakeEvalParams(const Problem& P, const Sensor& s, int64_t i) {
  EvalParams ep;
  ep.k = P.order;
  const int ki = s.seg[i] + P.order - 1;  // bspline.hpp:157-161
  ep.knot0 = P.knots[ki]; ep.knot1 = P.knots[ki + 1];
  ep.stamp = s.stamps[i];
  ep.basis = &P.basis[size_t(s.seg[i]) * P.order * P.order];
  ep.information = s.information;
  return ep;
}

template <class T>
static bool EvalFunctor(const Sensor& s, const EvalParams& ep, const double* meas, T const* const* params, T* r) {
  switch (s.kind) {
    case CALICO_SENSOR_CAMERA: return CameraResidual<T>(s.model, ep, meas, params, r);
    case CALICO_SENSOR_GYROSCOPE: return GyroscopeResidual<T>(s.model, ep, meas, params, r);
    default: return AccelerometerResidual<T>(s.model, ep, meas, params, r);
  }
}

// [Ceres] loss_function.cc: HuberLoss / CauchyLoss.
static void LossEvaluate(int loss, double a, double s, double rho[3]) {
  if (loss == CALICO_LOSS_HUBER) {
    const double b = a * a;
    if (s > b) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  } else if (loss == CALICO_LOSS_CAUCHY) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = b * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c * (inv * inv);
  } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

// [Ceres] manifold.h EigenQuaternionManifold::PlusJacobian, 4×3 row-major, storage x,y,z,w.
static void QuatPlusJacobian(const double* x, double J[12]) {
  const double X = x[0], Y = x[1], Z = x[2], W = x[3];
  J[0] = W;  J[1] = Z;  J[2] = -Y;
  J[3] = -Z; J[4] = W;  J[5] = X;
  J[6] = Y;  J[7] = -X; J[8] = W;
  J[9] = -X; J[10] = -Y; J[11] = -Z;
}
// [Ceres] EigenQuaternionManifold::Plus.
static void QuatPlus(const double* x, const double* delta, double* out) {
  const double nd = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (nd == 0.0) { for (int i = 0; i < 4; ++i) out[i] = x[i]; return; }
  const double sd = std::sin(nd) / nd;
  const Quat<double> qd(std::cos(nd), sd * delta[0], sd * delta[1], sd * delta[2]);
  const Quat<double> q = qd * Quat<double>::FromCoeffs(x);
  out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
}

// Build the reduced problem: [Ceres] the preprocessor drops constant and
// unused parameter blocks. Tangent order: control points (spline order), then
// the remaining blocks in block-id order.
static void BuildReduced(Problem& P) {
  for (auto& b : P.blocks) { b.used = false; b.tangent_offset = -1; }
  P.rblocks.clear(); P.res_offset.clear(); P.n_res = 0;
  for (size_t si = 0; si < P.sensors.size(); ++si) {
    const Sensor& s = P.sensors[si];
    for (int64_t i = 0; i < s.n(); ++i) {
      int ids[32]; const int n = BlockParams(P, s, i, ids);
      for (int j = 0; j < n; ++j) P.blocks[ids[j]].used = true;
      P.rblocks.push_back({int(si), i});
      P.res_offset.push_back(P.n_res);
      P.n_res += s.dim();
    }
  }
  {  // shard windows over spline segments (cal::shard_windows)
    const int nseg = int(P.valid_knots.size()) - 1;
    std::vector<int64_t> per_seg(size_t(std::max(nseg, 0)), 0);
    for (const BlockRef& b : P.rblocks) per_seg[size_t(P.sensors[b.sensor].seg[b.obs])] += 1;
    const std::vector<int> win = cal::shard_windows(per_seg, P.world);
    P.own.assign(P.rblocks.size(), 0);
    for (size_t b = 0; b < P.rblocks.size(); ++b) {
      const int sg = P.sensors[P.rblocks[b].sensor].seg[P.rblocks[b].obs];
      P.own[b] = (sg >= win[size_t(P.rank)] && sg < win[size_t(P.rank) + 1]) ? 1 : 0;
    }
  }
  P.reduced_blocks.clear();
  int off = 0;
  std::vector<char> is_ctrl(P.blocks.size(), 0);
  for (int id : P.ctrl) {
    is_ctrl[id] = 1;
    ParamBlock& b = P.blocks[id];
    if (!b.constant && b.used) { b.tangent_offset = off; off += b.tangent_size(); P.reduced_blocks.push_back(id); }
  }
  for (size_t id = 0; id < P.blocks.size(); ++id) {
    ParamBlock& b = P.blocks[id];
    if (is_ctrl[id] || b.constant || !b.used) continue;
    b.tangent_offset = off; off += b.tangent_size(); P.reduced_blocks.push_back(int(id));
  }
  P.n_eff = off;
}

struct Evaluation {
  bool ok = true;
  double cost = 0;
  std::vector<double> residuals;               // corrected
  std::vector<double> jac;                     // per block: dim × ncols (row-major), concatenated
  std::vector<int64_t> jac_offset;             // per block
  std::vector<int> col_start, col_ptr;         // per block: list of (tangent_offset, tangent_size) runs
  std::vector<int> col_off, col_size;
  std::vector<double> gradient;                // n_eff
};

// One residual block: residuals, and (if jac != nullptr) the local Jacobian
// dim × ncols with columns = concatenated tangent dims of the active blocks,
// both after loss correction. [Ceres] ResidualBlock::Evaluate +
// DynamicAutoDiffCostFunction::Evaluate (stride 4) + Corrector.
static bool EvaluateBlock(const Problem& P, const std::vector<const double*>& xptr, const Sensor& s, int64_t i,
                          bool apply_loss, double* r, double* jac, int* ncols_out, int* offs, int* sizes, int* nruns,
                          double* cost) {
  int ids[32]; const int np = BlockParams(P, s, i, ids);
  const EvalParams ep = MakeEvalParams(P, s, i);
  const int dim = s.dim();
  const double* meas = &s.meas[size_t(i) * dim];
  const double* pd[32];
  for (int j = 0; j < np; ++j) pd[j] = xptr[ids[j]];
  if (!jac) {
    if (!EvalFunctor<double>(s, ep, meas, pd, r)) return false;
  } else {
    // ambient Jacobian by 4-wide passes over the active (non-constant) parameters
    constexpr int S = 4;
    typedef Dual<S> D;
    int amb_start[32], amb_total = 0, total_amb_all = 0;
    D storage[128];
    const D* pj[32];
    int pos[32];
    for (int j = 0; j < np; ++j) { pos[j] = total_amb_all; total_amb_all += P.blocks[ids[j]].size; }
    for (int j = 0; j < np; ++j) {
      const ParamBlock& b = P.blocks[ids[j]];
      amb_start[j] = (b.tangent_offset >= 0) ? amb_total : -1;
      if (b.tangent_offset >= 0) amb_total += b.size;
      pj[j] = &storage[pos[j]];
    }
    double Jamb[3 * 128];
    const int npass = (amb_total + S - 1) / S;
    bool first = true;
    for (int pass = 0; pass < std::max(npass, 1); ++pass) {
      for (int j = 0; j < np; ++j) {
        const ParamBlock& b = P.blocks[ids[j]];
        for (int c = 0; c < b.size; ++c) {
          D& d = storage[pos[j] + c];
          d = D(pd[j][c]);
          if (amb_start[j] >= 0) {
            const int g = amb_start[j] + c - pass * S;
            if (g >= 0 && g < S) d.d[g] = 1.0;
          }
        }
      }
      D rr[3];
      if (!EvalFunctor<D>(s, ep, meas, pj, rr)) return false;
      if (first) { for (int k = 0; k < dim; ++k) r[k] = rr[k].v; first = false; }
      for (int g = 0; g < S; ++g) {
        const int col = pass * S + g;
        if (col >= amb_total) break;
        for (int k = 0; k < dim; ++k) Jamb[k * amb_total + col] = rr[k].d[g];
      }
    }
    // project to the tangent space
    int ncols = 0, nr = 0;
    for (int j = 0; j < np; ++j) {
      const ParamBlock& b = P.blocks[ids[j]];
      if (b.tangent_offset < 0) continue;
      offs[nr] = b.tangent_offset; sizes[nr] = b.tangent_size(); ++nr;
      ncols += b.tangent_size();
    }
    int c0 = 0;
    for (int j = 0; j < np; ++j) {
      const ParamBlock& b = P.blocks[ids[j]];
      if (b.tangent_offset < 0) continue;
      if (b.manifold == CALICO_MANIFOLD_EIGEN_QUATERNION) {
        double PJ[12]; QuatPlusJacobian(pd[j], PJ);
        for (int k = 0; k < dim; ++k)
          for (int c = 0; c < 3; ++c) {
            double sacc = 0;
            for (int a = 0; a < 4; ++a) sacc += Jamb[k * amb_total + amb_start[j] + a] * PJ[a * 3 + c];
            jac[k * ncols + c0 + c] = sacc;
          }
        c0 += 3;
      } else {
        for (int k = 0; k < dim; ++k)
          for (int c = 0; c < b.size; ++c) jac[k * ncols + c0 + c] = Jamb[k * amb_total + amb_start[j] + c];
        c0 += b.size;
      }
    }
    *ncols_out = ncols; *nruns = nr;
  }
  double sq = 0; for (int k = 0; k < dim; ++k) sq += r[k] * r[k];
  if (!apply_loss || s.loss == CALICO_LOSS_NONE) { *cost = 0.5 * sq; return true; }
  double rho[3]; LossEvaluate(s.loss, s.loss_scale, sq, rho);
  *cost = 0.5 * rho[0];
  // [Ceres] corrector.cc
  const double sqrt_rho1 = std::sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
  else {
    const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(Dd);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / sq;
  }
  if (jac) {
    const int ncols = *ncols_out;
    if (alpha_sq_norm == 0.0) { for (int q = 0; q < dim * ncols; ++q) jac[q] *= sqrt_rho1; }
    else {
      for (int c = 0; c < ncols; ++c) {
        double rtj = 0; for (int k = 0; k < dim; ++k) rtj += jac[k * ncols + c] * r[k];
        for (int k = 0; k < dim; ++k) jac[k * ncols + c] = sqrt_rho1 * (jac[k * ncols + c] - alpha_sq_norm * r[k] * rtj);
      }
    }
  }
  for (int k = 0; k < dim; ++k) r[k] *= residual_scaling;
  return true;
}

static std::vector<const double*> XPointers(const Problem& P, const std::vector<std::vector<double>>* override_vals) {
  std::vector<const double*> x(P.blocks.size());
  for (size_t i = 0; i < P.blocks.size(); ++i) x[i] = override_vals ? (*override_vals)[i].data() : P.blocks[i].v.data();
  return x;
}

// [Ceres] ProgramEvaluator::Evaluate. vals: parameter values per block.
static void Evaluate(const Problem& P, const std::vector<std::vector<double>>& vals, bool want_jac, int num_threads,
                     Evaluation* E) {
  const int64_t nb = int64_t(P.rblocks.size());
  const std::vector<const double*> x = XPointers(P, &vals);
  E->ok = true; E->cost = 0;
  E->residuals.assign(size_t(P.n_res), 0.0);
  if (want_jac) {
    // layout pass
    E->jac_offset.assign(nb + 1, 0); E->col_ptr.assign(nb + 1, 0);
    E->col_off.clear(); E->col_size.clear();
    int64_t jo = 0;
    for (int64_t b = 0; b < nb; ++b) {
      const Sensor& s = P.sensors[P.rblocks[b].sensor];
      int ids[32]; const int np = BlockParams(P, s, P.rblocks[b].obs, ids);
      int ncols = 0;
      E->col_ptr[b] = int(E->col_off.size());
      for (int j = 0; j < np; ++j) {
        const ParamBlock& pb = P.blocks[ids[j]];
        if (pb.tangent_offset < 0) continue;
        E->col_off.push_back(pb.tangent_offset); E->col_size.push_back(pb.tangent_size());
        ncols += pb.tangent_size();
      }
      E->jac_offset[b] = jo; jo += int64_t(s.dim()) * ncols;
    }
    E->col_ptr[nb] = int(E->col_off.size());
    E->jac_offset[nb] = jo;
    E->jac.assign(size_t(jo), 0.0);
    E->gradient.assign(size_t(P.n_eff), 0.0);
  }
  const int T = std::max(1, num_threads);
  std::vector<double> tcost(T, 0.0);
  std::vector<char> tok(T, 1);
  auto work = [&](int tid) {
    const int64_t lo = nb * tid / T, hi = nb * (tid + 1) / T;
    int offs[32], sizes[32], nruns, ncols;
    double c;
    for (int64_t b = lo; b < hi; ++b) {
      if (!P.own[size_t(b)]) continue;
      const Sensor& s = P.sensors[P.rblocks[b].sensor];
      double* r = &E->residuals[size_t(P.res_offset[b])];
      double* jac = want_jac ? &E->jac[size_t(E->jac_offset[b])] : nullptr;
      if (!EvaluateBlock(P, x, s, P.rblocks[b].obs, true, r, jac, &ncols, offs, sizes, &nruns, &c)) { tok[tid] = 0; continue; }
      tcost[tid] += c;
    }
  };
  if (T == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& t : th) t.join(); }
  for (int t = 0; t < T; ++t) { E->cost += tcost[t]; if (!tok[t]) E->ok = false; }
  {
    double cv[2] = {E->cost, E->ok ? 0.0 : 1.0};
    P.reduce(cv, 2);
    E->cost = cv[0]; E->ok = cv[1] == 0.0;
  }
  if (want_jac && E->ok) {
    for (int64_t b = 0; b < nb; ++b) {
      const Sensor& s = P.sensors[P.rblocks[b].sensor];
      const int dim = s.dim();
      const double* r = &E->residuals[size_t(P.res_offset[b])];
      const double* J = &E->jac[size_t(E->jac_offset[b])];
      int ncols = 0; for (int q = E->col_ptr[b]; q < E->col_ptr[b + 1]; ++q) ncols += E->col_size[q];
      int c0 = 0;
      for (int q = E->col_ptr[b]; q < E->col_ptr[b + 1]; ++q) {
        for (int c = 0; c < E->col_size[q]; ++c) {
          double g = 0; for (int k = 0; k < dim; ++k) g += J[k * ncols + c0 + c] * r[k];
          E->gradient[E->col_off[q] + c] += g;
        }
        c0 += E->col_size[q];
      }
    }
    P.reduce(E->gradient.data(), int64_t(E->gradient.size()));
  }
}

// Dense JᵀJ (n_eff × n_eff, row-major, full symmetric) with optional column scaling.
static void AccumulateJtJ(const Problem& P, const Evaluation& E, const double* scale, int num_threads,
                          std::vector<double>* H) {
  const int n = P.n_eff;
  const int64_t nb = int64_t(P.rblocks.size());
  // one dense copy per thread: at most 16, and no more than ~2 GB of them
  const int T = std::max(1, std::min(std::min(num_threads, 16), int(2.5e8 / (double(n) * n + 1.0))));
  std::vector<std::vector<double>> Ht(T);
  auto work = [&](int tid) {
    std::vector<double>& h = Ht[tid];
    h.assign(size_t(n) * n, 0.0);
    const int64_t lo = nb * tid / T, hi = nb * (tid + 1) / T;
    int gcol[128]; double sc[128];
    for (int64_t b = lo; b < hi; ++b) {
      const Sensor& s = P.sensors[P.rblocks[b].sensor];
      const int dim = s.dim();
      const double* J = &E.jac[size_t(E.jac_offset[b])];
      int ncols = 0;
      for (int q = E.col_ptr[b]; q < E.col_ptr[b + 1]; ++q)
        for (int c = 0; c < E.col_size[q]; ++c) { gcol[ncols] = E.col_off[q] + c; sc[ncols] = scale ? scale[gcol[ncols]] : 1.0; ++ncols; }
      for (int a = 0; a < ncols; ++a)
        for (int c = a; c < ncols; ++c) {
          double v = 0; for (int k = 0; k < dim; ++k) v += (J[k * ncols + a] * sc[a]) * (J[k * ncols + c] * sc[c]);
          const int ga = gcol[a], gc = gcol[c];
          if (ga <= gc) h[size_t(ga) * n + gc] += v; else h[size_t(gc) * n + ga] += v;
        }
    }
  };
  if (T == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& t : th) t.join(); }
  H->assign(size_t(n) * n, 0.0);
  for (int t = 0; t < T; ++t) for (size_t q = 0; q < H->size(); ++q) (*H)[q] += Ht[t][q];
  P.reduce(H->data(), int64_t(H->size()));
  for (int a = 0; a < n; ++a) for (int c = a + 1; c < n; ++c) (*H)[size_t(c) * n + a] = (*H)[size_t(a) * n + c];
}

// Dense Cholesky solve (lower). Returns false if not positive definite.
static bool CholeskySolve(std::vector<double>& A, int n, std::vector<double>& b) {
  for (int j = 0; j < n; ++j) {
    double d = A[size_t(j) * n + j];
    for (int k = 0; k < j; ++k) d -= A[size_t(j) * n + k] * A[size_t(j) * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[size_t(j) * n + j] = d;
    const double inv = 1.0 / d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[size_t(i) * n + j];
      const double* ri = &A[size_t(i) * n];
      const double* rj = &A[size_t(j) * n];
      for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
      A[size_t(i) * n + j] = s * inv;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[size_t(i) * n + k] * b[k];
    b[i] = s / A[size_t(i) * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= A[size_t(k) * n + i] * b[k];
    b[i] = s / A[size_t(i) * n + i];
  }
  return true;
}

// The same factorisation for large systems (long trajectories: thousands of unknowns), blocked by 64 columns with the
// panel solve and the trailing update spread over threads. Only the order of the additions differs from the loop above.
static bool CholeskySolveBlocked(std::vector<double>& A, int n, std::vector<double>& b, int num_threads) {
  const int NB = 64;
  const int T = std::max(1, num_threads);
  bool ok = true;
  auto parallel_rows = [&](int lo, int hi, auto&& fn) {      // rows dealt cyclically: the work per row grows with the row
    if (T == 1 || hi - lo < 4 * T) { for (int i = lo; i < hi; ++i) fn(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (int i = lo + t; i < hi; i += T) fn(i); });
    for (auto& x : th) x.join();
  };
  for (int j0 = 0; j0 < n && ok; j0 += NB) {
    const int jb = std::min(NB, n - j0), j1 = j0 + jb;
    for (int j = j0; j < j1; ++j) {                             // diagonal block
      double d = A[size_t(j) * n + j];
      for (int k = j0; k < j; ++k) d -= A[size_t(j) * n + k] * A[size_t(j) * n + k];
      if (!(d > 0.0) || !std::isfinite(d)) { ok = false; break; }
      d = std::sqrt(d);
      A[size_t(j) * n + j] = d;
      for (int i = j + 1; i < j1; ++i) {
        double v = A[size_t(i) * n + j];
        for (int k = j0; k < j; ++k) v -= A[size_t(i) * n + k] * A[size_t(j) * n + k];
        A[size_t(i) * n + j] = v / d;
      }
    }
    if (!ok) break;
    parallel_rows(j1, n, [&](int i) {                           // panel: L(i, j0:j1) = A(i, j0:j1) L_jj^-T
      double* ri = &A[size_t(i) * n];
      for (int j = j0; j < j1; ++j) {
        const double* rj = &A[size_t(j) * n];
        double v = ri[j];
        for (int k = j0; k < j; ++k) v -= ri[k] * rj[k];
        ri[j] = v / rj[j];
      }
    });
    parallel_rows(j1, n, [&](int i) {                           // trailing update of the lower triangle
      double* ri = &A[size_t(i) * n];
      const double* li = ri + j0;
      for (int c = j1; c <= i; ++c) {
        const double* lc = &A[size_t(c) * n + j0];
        double v = 0;
        for (int k = 0; k < jb; ++k) v += li[k] * lc[k];
        ri[c] -= v;
      }
    });
  }
  if (!ok) return false;
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= A[size_t(i) * n + k] * b[k];
    b[i] = v / A[size_t(i) * n + i];
  }
  // backward sweep in axpy form: the factor is stored by rows, a column walk would miss the cache on every element
  for (int i = n - 1; i >= 0; --i) {
    b[i] /= A[size_t(i) * n + i];
    const double bi = b[i];
    const double* ri = &A[size_t(i) * n];
    for (int k = 0; k < i; ++k) b[k] -= ri[k] * bi;
  }
  return true;
}

// [Ceres] ProgramEvaluator::Plus over the reduced blocks.
static void PlusAll(const Problem& P, const std::vector<std::vector<double>>& x, const double* delta,
                    std::vector<std::vector<double>>* out) {
  *out = x;
  for (int id : P.reduced_blocks) {
    const ParamBlock& b = P.blocks[id];
    const double* d = delta + b.tangent_offset;
    if (b.manifold == CALICO_MANIFOLD_EIGEN_QUATERNION) QuatPlus(x[id].data(), d, (*out)[id].data());
    else for (int c = 0; c < b.size; ++c) (*out)[id][c] = x[id][c] + d[c];
  }
}
static double ReducedNorm(const Problem& P, const std::vector<std::vector<double>>& x) {
  double s = 0;
  for (int id : P.reduced_blocks) for (double v : x[id]) s += v * v;
  return std::sqrt(s);
}
static void ReducedDiffNorms(const Problem& P, const std::vector<std::vector<double>>& a,
                             const std::vector<std::vector<double>>& b, double* l2, double* linf) {
  double s = 0, m = 0;
  for (int id : P.reduced_blocks)
    for (size_t c = 0; c < a[id].size(); ++c) { const double d = a[id][c] - b[id][c]; s += d * d; m = std::max(m, std::fabs(d)); }
  *l2 = std::sqrt(s); *linf = m;
}

static void FillCounts(const Problem& P, calico_summary* sm) {
  sm->num_residual_blocks = int(P.rblocks.size());
  sm->num_residuals = int(P.n_res);
  sm->num_parameter_blocks = int(P.blocks.size());
  int np = 0, ne = 0;
  for (const auto& b : P.blocks) { np += b.size; ne += b.tangent_size(); }
  sm->num_parameters = np; sm->num_effective_parameters = ne;
  sm->num_residual_blocks_reduced = sm->num_residual_blocks;
  sm->num_residuals_reduced = sm->num_residuals;
  sm->num_parameter_blocks_reduced = int(P.reduced_blocks.size());
  int npr = 0;
  for (int id : P.reduced_blocks) npr += P.blocks[id].size;
  sm->num_parameters_reduced = npr;
  sm->num_effective_parameters_reduced = P.n_eff;
}

// [Ceres] TrustRegionMinimizer::Minimize with LevenbergMarquardtStrategy and a
// dense normal-equation solve (DENSE_SCHUR solves the same system exactly).
static int Solve(Problem& P, const calico_solver_options& o, calico_summary* sm) {
  const auto t_start = std::chrono::steady_clock::now();
  std::memset(sm, 0, sizeof(*sm));
  if (P.order <= 0) return P.set_error(CALICO_FAILED_PRECONDITION, "spline not set");
  BuildReduced(P);
  FillCounts(P, sm);
  P.iterations.clear();
  const int n = P.n_eff;
  std::vector<std::vector<double>> x(P.blocks.size()), cand;
  for (size_t i = 0; i < P.blocks.size(); ++i) x[i] = P.blocks[i].v;
  Evaluation E, Ec;
  P.t_evaluate = P.t_assemble = P.t_linear_solve = 0;
  auto timed = [](double& acc, auto&& fn) {
    const auto t0 = std::chrono::steady_clock::now();
    fn();
    acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  auto finish = [&](int term, const char* msg) {
    sm->termination_type = term;
    std::snprintf(sm->message, sizeof(sm->message), "%s", msg);
    if (term != CALICO_FAILURE) {
      sm->final_cost = sm->initial_cost;
      for (const auto& it : P.iterations) sm->final_cost = std::min(sm->final_cost, it.cost);
    }
    sm->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    sm->solve_time_in_seconds = sm->total_time_in_seconds;
    return CALICO_OK;
  };
  // iteration 0
  double x_norm = ReducedNorm(P, x);
  timed(P.t_evaluate, [&] { Evaluate(P, x, true, o.num_threads, &E); });
  sm->num_jacobian_evaluations++;
  if (!E.ok) return finish(CALICO_FAILURE, "Initial residual and Jacobian evaluation failed.");
  double x_cost = E.cost;
  sm->initial_cost = x_cost;
  std::vector<double> scale(n, 1.0);
  std::vector<double> H;
  if (o.jacobi_scaling) {
    std::vector<double> Hd;
    timed(P.t_assemble, [&] { AccumulateJtJ(P, E, nullptr, o.num_threads, &Hd); });
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(Hd[size_t(i) * n + i]));
  }
  timed(P.t_assemble, [&] { AccumulateJtJ(P, E, scale.data(), o.num_threads, &H); });
  auto gradient_norms = [&](double* gmax, double* gnorm) {
    // |x - Plus(x, -g)|, [Ceres] trust_region_minimizer.cc EvaluateGradientAndJacobian
    std::vector<double> ng(n);
    for (int i = 0; i < n; ++i) ng[i] = -E.gradient[i];
    std::vector<std::vector<double>> xp;
    PlusAll(P, x, ng.data(), &xp);
    ReducedDiffNorms(P, x, xp, gnorm, gmax);
  };
  TrustRegionControl tr{o.initial_trust_region_radius, 2.0};
  double& radius = tr.radius;
  bool reuse_diagonal = false;
  std::vector<double> diagonal(n, 0.0);
  calico_iteration it{};
  it.iteration = 0; it.cost = x_cost; it.trust_region_radius = radius;
  double gnorm;
  gradient_norms(&it.gradient_max_norm, &gnorm);
  int num_consecutive_invalid = 0;
  bool printed_header = false;
  auto log_row = [&](const calico_iteration& r) {
    if (!o.minimizer_progress_to_stdout) return;
    if (!printed_header) { std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n"); printed_header = true; }
    std::printf("%4d % 8e   % 3.2e   % 3.2e  % 3.2e  % 3.2e % 3.2e\n", r.iteration, r.cost, r.cost_change,
                r.gradient_max_norm, r.step_norm, r.relative_decrease, r.trust_region_radius);
  };
  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.iteration > 0) { if (it.step_is_successful) sm->num_successful_steps++; else sm->num_unsuccessful_steps++; }
    it.trust_region_radius = radius;
    P.iterations.push_back(it);
    log_row(it);
    sm->num_iterations = it.iteration;
    if (it.iteration >= o.max_num_iterations) { finish(CALICO_NO_CONVERGENCE, "Maximum number of iterations reached."); break; }
    if (it.gradient_max_norm <= o.gradient_tolerance) { finish(CALICO_CONVERGENCE, "Gradient tolerance reached."); break; }
    if (radius < o.min_trust_region_radius) { finish(CALICO_CONVERGENCE, "Minimum trust region radius reached."); break; }
    const double prev_gmax = it.gradient_max_norm;
    calico_iteration nit{};
    nit.iteration = it.iteration + 1;
    it = nit;
    // ComputeTrustRegionStep: [Ceres] LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal) {
      for (int i = 0; i < n; ++i)
        diagonal[i] = std::min(std::max(H[size_t(i) * n + i], o.min_lm_diagonal), o.max_lm_diagonal);
    }
    std::vector<double> A = H;
    std::vector<double> gs(n);
    for (int i = 0; i < n; ++i) { A[size_t(i) * n + i] += diagonal[i] / radius; gs[i] = E.gradient[i] * scale[i]; }
    std::vector<double> y = gs;
    bool solved = false;
    timed(P.t_linear_solve, [&] { solved = n > 1500 ? CholeskySolveBlocked(A, n, y, o.num_threads) : CholeskySolve(A, n, y); });
    if (solved) for (int i = 0; i < n; ++i) if (!std::isfinite(y[i])) solved = false;
    reuse_diagonal = true;
    double model_cost_change = 0;
    std::vector<double> step(n), delta(n);
    if (solved) {
      for (int i = 0; i < n; ++i) step[i] = -y[i];
      // model_cost_change = -(J step)'(f + J step / 2), J = scaled Jacobian
      const int64_t nb = int64_t(P.rblocks.size());
      double mcc = 0;
      for (int64_t b = 0; b < nb; ++b) {
        const Sensor& s = P.sensors[P.rblocks[b].sensor];
        const int dim = s.dim();
        const double* J = &E.jac[size_t(E.jac_offset[b])];
        const double* r = &E.residuals[size_t(P.res_offset[b])];
        int ncols = 0; for (int q = E.col_ptr[b]; q < E.col_ptr[b + 1]; ++q) ncols += E.col_size[q];
        for (int k = 0; k < dim; ++k) {
          double m = 0; int c0 = 0;
          for (int q = E.col_ptr[b]; q < E.col_ptr[b + 1]; ++q) {
            for (int c = 0; c < E.col_size[q]; ++c) { const int g = E.col_off[q] + c; m += J[k * ncols + c0 + c] * scale[g] * step[g]; }
            c0 += E.col_size[q];
          }
          mcc += m * (r[k] + m / 2.0);
        }
      }
      P.reduce(&mcc, 1);
      model_cost_change = -mcc;
      it.step_is_valid = model_cost_change > 0.0;
    }
    if (!it.step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid >= o.max_num_consecutive_invalid_steps) {
        finish(CALICO_FAILURE, "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.");
        break;
      }
      tr.step_is_invalid(); reuse_diagonal = true;
      it.cost = x_cost; it.cost_change = 0; it.gradient_max_norm = prev_gmax; it.step_norm = 0; it.relative_decrease = 0;
      continue;
    }
    num_consecutive_invalid = 0;
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    // ComputeCandidatePointAndEvaluateCost
    PlusAll(P, x, delta.data(), &cand);
    timed(P.t_evaluate, [&] { Evaluate(P, cand, false, o.num_threads, &Ec); });
    sm->num_cost_evaluations++;
    double candidate_cost = Ec.ok ? Ec.cost : std::numeric_limits<double>::max();
    // ParameterToleranceReached
    double step_inf;
    ReducedDiffNorms(P, x, cand, &it.step_norm, &step_inf);
    if (it.step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
      finish(CALICO_CONVERGENCE, "Parameter tolerance reached."); break;
    }
    // FunctionToleranceReached
    it.cost_change = x_cost - candidate_cost;
    if (std::fabs(it.cost_change) <= o.function_tolerance * x_cost) {
      finish(CALICO_CONVERGENCE, "Function tolerance reached."); break;
    }
    // IsStepSuccessful
    it.relative_decrease = TrustRegionControl::relative_decrease(x_cost, candidate_cost, model_cost_change);
    if (it.relative_decrease > o.min_relative_decrease) {
      // HandleSuccessfulStep
      x = cand; x_norm = ReducedNorm(P, x);
      timed(P.t_evaluate, [&] { Evaluate(P, x, true, o.num_threads, &E); });
      sm->num_jacobian_evaluations++;
      if (!E.ok) { finish(CALICO_FAILURE, "Residual and Jacobian evaluation failed."); break; }
      x_cost = E.cost;
      timed(P.t_assemble, [&] { AccumulateJtJ(P, E, scale.data(), o.num_threads, &H); });
      it.cost = x_cost; it.step_is_successful = 1;
      gradient_norms(&it.gradient_max_norm, &gnorm);
      tr.step_accepted(it.relative_decrease, o.max_trust_region_radius); reuse_diagonal = false;
    } else {
      it.step_is_successful = 0; it.cost = candidate_cost; it.gradient_max_norm = prev_gmax;
      tr.step_rejected(); reuse_diagonal = true;
    }
  }
  // write back the best point (monotonic steps: the last accepted x)
  for (size_t i = 0; i < P.blocks.size(); ++i) P.blocks[i].v = x[i];
  return CALICO_OK;
}

}  // namespace oracle

using oracle::Problem;

// ---------------------------------------------------------------------------
// C API: same shape as include/calico_hip.h, prefix oracle_.
// ---------------------------------------------------------------------------
extern "C" {

int32_t oracle_problem_create(Problem** out) { *out = new Problem(); return CALICO_OK; }

// Test hook: the trust-region control driven by a given sequence of step qualities. Row i: candidate cost =
// x_cost - rho[i] * model_cost_change with x_cost = model_cost_change = 1, or "cannot be evaluated" when infinite[i];
// out: radius after the row, whether the step was accepted, and what the cost column of the iteration log shows.
int32_t oracle_lm_control_replay(int32_t n, const double* rho, const int32_t* infinite, double initial_radius,
                                 double min_relative_decrease, double max_radius, double* radius_out,
                                 int32_t* accepted_out, double* cost_column_out) {
  oracle::TrustRegionControl tr{initial_radius, 2.0};
  const double x_cost = 1.0, mcc = 1.0;
  for (int32_t i = 0; i < n; ++i) {
    const double candidate_cost = infinite[i] ? std::numeric_limits<double>::max() : x_cost - rho[i] * mcc;
    const double r = oracle::TrustRegionControl::relative_decrease(x_cost, candidate_cost, mcc);
    const bool ok = r > min_relative_decrease;
    if (ok) tr.step_accepted(r, max_radius); else tr.step_rejected();
    radius_out[i] = tr.radius; accepted_out[i] = ok ? 1 : 0;
    cost_column_out[i] = candidate_cost;      // accepted: the new x_cost (= the candidate's); rejected: the candidate's cost
  }
  return CALICO_OK;
}
void oracle_problem_destroy(Problem* p) { delete p; }
const char* oracle_last_error(const Problem* p) { return p->error.c_str(); }

void oracle_default_solver_options(calico_solver_options* o) {
  // batch_optimizer.cpp:10-17 + [Ceres] Solver::Options defaults
  o->max_num_iterations = 50; o->num_threads = 1; o->minimizer_progress_to_stdout = 1; o->jacobi_scaling = 1;
  o->max_num_consecutive_invalid_steps = 5; o->sync_every = 1;
  o->function_tolerance = 1e-8; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-10;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
}

int32_t oracle_problem_add_param_block(Problem* p, const double* values, int32_t size, int32_t manifold,
                                       int32_t is_constant, int32_t* id_out) {
  if (size <= 0 || !values) return p->set_error(CALICO_INVALID_ARGUMENT, "bad parameter block");
  if (manifold == CALICO_MANIFOLD_EIGEN_QUATERNION && size != 4)
    return p->set_error(CALICO_INVALID_ARGUMENT, "quaternion manifold needs size 4");
  oracle::ParamBlock b; b.v.assign(values, values + size); b.size = size; b.manifold = manifold; b.constant = is_constant != 0;
  p->blocks.push_back(b);
  if (id_out) *id_out = int32_t(p->blocks.size()) - 1;
  return CALICO_OK;
}
int32_t oracle_get_param_block(Problem* p, int32_t id, double* out) {
  if (id < 0 || id >= int(p->blocks.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block id");
  std::copy(p->blocks[id].v.begin(), p->blocks[id].v.end(), out); return CALICO_OK;
}
int32_t oracle_set_param_block(Problem* p, int32_t id, const double* v) {
  if (id < 0 || id >= int(p->blocks.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block id");
  std::copy(v, v + p->blocks[id].size, p->blocks[id].v.begin()); return CALICO_OK;
}
int32_t oracle_get_param_blocks(Problem* p, int32_t n, const int32_t* ids, double* out) {
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= int(p->blocks.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block id");
    std::copy(p->blocks[ids[i]].v.begin(), p->blocks[ids[i]].v.end(), out);
    out += p->blocks[ids[i]].size;
  }
  return CALICO_OK;
}
int32_t oracle_set_param_blocks(Problem* p, int32_t n, const int32_t* ids, const double* v) {
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= int(p->blocks.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block id");
    std::copy(v, v + p->blocks[ids[i]].size, p->blocks[ids[i]].v.begin());
    v += p->blocks[ids[i]].size;
  }
  return CALICO_OK;
}
int32_t oracle_problem_set_spline(Problem* p, int32_t order, int32_t n_knots, const double* knots, const double* basis,
                                  const int32_t* ctrl) {
  if (order < 2 || order > 16 || n_knots < 2 * order) return p->set_error(CALICO_INVALID_ARGUMENT, "bad spline");
  p->order = order; p->knots.assign(knots, knots + n_knots);
  const int deg = order - 1;
  p->valid_knots.assign(knots + deg, knots + n_knots - deg);
  const int nseg = int(p->valid_knots.size()) - 1;
  p->basis.assign(basis, basis + size_t(nseg) * order * order);
  p->ctrl.assign(ctrl, ctrl + (n_knots - order));
  for (int id : p->ctrl)
    if (id < 0 || id >= int(p->blocks.size()) || p->blocks[id].size != 6)
      return p->set_error(CALICO_INVALID_ARGUMENT, "control point blocks must be 6-vectors");
  return CALICO_OK;
}
int32_t oracle_problem_add_rigid_body(Problem* p, int32_t q, int32_t t, int32_t* id_out) {
  p->bodies.push_back({q, t}); if (id_out) *id_out = int32_t(p->bodies.size()) - 1; return CALICO_OK;
}
int32_t oracle_problem_add_sensor(Problem* p, int32_t kind, int32_t model, int32_t intr, int32_t q, int32_t t, int32_t lat,
                                  int32_t grav, double sigma, int32_t loss, double loss_scale, int32_t* id_out) {
  const int K = kind == CALICO_SENSOR_CAMERA ? oracle::CameraNumParams(model) : oracle::ImuNumParams(model);
  if (K < 0) return p->set_error(CALICO_FAILED_PRECONDITION, "sensor model is not defined");
  if (intr < 0 || intr >= int(p->blocks.size()) || p->blocks[intr].size != K)
    return p->set_error(CALICO_INVALID_ARGUMENT, "intrinsics block size does not match the model");
  oracle::Sensor s; s.kind = kind; s.model = model; s.intr = intr; s.q = q; s.t = t; s.lat = lat; s.grav = grav;
  s.sigma = sigma; s.information = sigma > 0.0 ? 1.0 / sigma : 1.0;  // camera_cost_functor.cpp:15
  s.loss = loss; s.loss_scale = loss_scale;
  p->sensors.push_back(s); if (id_out) *id_out = int32_t(p->sensors.size()) - 1; return CALICO_OK;
}
static int32_t add_obs(Problem* p, int32_t sid, int64_t n, const double* meas, const double* stamps, const int32_t* body,
                       const int32_t* point) {
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  if (p->order <= 0) return p->set_error(CALICO_FAILED_PRECONDITION, "spline must be set before residuals");
  oracle::Sensor& s = p->sensors[sid];
  const int dim = s.dim();
  for (int64_t i = 0; i < n; ++i) {
    const int seg = oracle::GetSplineIndex(*p, stamps[i]);
    if (seg < 0) return p->set_error(CALICO_INVALID_ARGUMENT, "measurement stamp is outside the spline's valid knots");
    if (body) {
      if (body[i] < 0 || body[i] >= int(p->bodies.size()))
        return p->set_error(CALICO_FAILED_PRECONDITION, "observation of a rigid body that does not exist in the world model");
      s.body.push_back(body[i]); s.point.push_back(point[i]);
    }
    s.seg.push_back(seg); s.stamps.push_back(stamps[i]);
    for (int k = 0; k < dim; ++k) s.meas.push_back(meas[i * dim + k]);
  }
  return CALICO_OK;
}
int32_t oracle_problem_add_camera_residuals(Problem* p, int32_t sid, int64_t n, const double* px, const double* st,
                                            const int32_t* body, const int32_t* point) {
  if (sid >= 0 && sid < int(p->sensors.size()) && p->sensors[sid].kind != CALICO_SENSOR_CAMERA)
    return p->set_error(CALICO_INVALID_ARGUMENT, "sensor is not a camera");
  return add_obs(p, sid, n, px, st, body, point);
}
int32_t oracle_problem_add_imu_residuals(Problem* p, int32_t sid, int64_t n, const double* m, const double* st) {
  if (sid >= 0 && sid < int(p->sensors.size()) && p->sensors[sid].kind == CALICO_SENSOR_CAMERA)
    return p->set_error(CALICO_INVALID_ARGUMENT, "sensor is not an IMU sensor");
  return add_obs(p, sid, n, m, st, nullptr, nullptr);
}
int32_t oracle_solve(Problem* p, const calico_solver_options* o, calico_summary* sm) { return oracle::Solve(*p, *o, sm); }
// Test hook: the blocked, threaded Cholesky (systems beyond 1500 unknowns) against the plain loop on the same SPD system
// A (n x n, row-major) x = b; returns the largest |x_blocked - x_plain| / max|x_plain|, or -1 when either fails.
double oracle_cholesky_blocked_vs_plain(int32_t n, const double* A, const double* b, int32_t num_threads) {
  std::vector<double> A1(A, A + size_t(n) * n), A2 = A1, b1(b, b + n), b2 = b1;
  if (!oracle::CholeskySolve(A1, n, b1) || !oracle::CholeskySolveBlocked(A2, n, b2, num_threads)) return -1.0;
  double d = 0, m = 0;
  for (int i = 0; i < n; ++i) { d = std::max(d, std::fabs(b1[i] - b2[i])); m = std::max(m, std::fabs(b1[i])); }
  return m > 0 ? d / m : d;
}
// seconds of the last oracle_solve by part: [residual/Jacobian evaluation, normal-equation assembly, dense factorisation + solve]
int32_t oracle_get_solve_timing(Problem* p, double* out3) {
  out3[0] = p->t_evaluate; out3[1] = p->t_assemble; out3[2] = p->t_linear_solve;
  return CALICO_OK;
}
int32_t oracle_problem_set_shard(Problem* p, int32_t rank, int32_t world) {
  if (world < 1 || rank < 0 || rank >= world) return p->set_error(CALICO_INVALID_ARGUMENT, "bad rank / world size");
  p->rank = rank; p->world = world; return CALICO_OK;
}
// host-buffer all-reduce (sum) callback: fn(ctx, buf, n)
int32_t oracle_problem_set_allreduce(Problem* p, int32_t (*fn)(void*, double*, int64_t), void* ctx) {
  p->allreduce = fn; p->allreduce_ctx = ctx; return CALICO_OK;
}
int32_t oracle_get_iterations(Problem* p, calico_iteration* out, int32_t max_rows, int32_t* n_out) {
  const int n = std::min<int>(max_rows, int(p->iterations.size()));
  for (int i = 0; i < n; ++i) out[i] = p->iterations[i];
  *n_out = n; return CALICO_OK;
}
// Sensor::UpdateResiduals: apply_loss_function = false (camera.cpp:70-80).
int32_t oracle_get_residuals(Problem* p, int32_t sid, double* out, uint8_t* valid) {
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  const oracle::Sensor& s = p->sensors[sid];
  const std::vector<const double*> x = oracle::XPointers(*p, nullptr);
  bool all = true;
  for (int64_t i = 0; i < s.n(); ++i) {
    double c; int a, b[32], c2[32], d;
    const bool ok = oracle::EvaluateBlock(*p, x, s, i, false, out + i * s.dim(), nullptr, &a, b, c2, &d, &c);
    if (valid) valid[i] = ok ? 1 : 0;
    if (!ok) { all = false; for (int k = 0; k < s.dim(); ++k) out[i * s.dim() + k] = 0; }
  }
  return all ? CALICO_OK : p->set_error(CALICO_INTERNAL, "Failed to update residual");
}
int32_t oracle_get_inlier_mask(Problem* p, int32_t sid, double thr, uint8_t* mask) {
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  const oracle::Sensor& s = p->sensors[sid];
  std::vector<double> r(size_t(s.n()) * s.dim()); std::vector<uint8_t> v(size_t(s.n()));
  oracle_get_residuals(p, sid, r.data(), v.data());
  for (int64_t i = 0; i < s.n(); ++i) {
    double sq = 0; for (int k = 0; k < s.dim(); ++k) sq += r[i * s.dim() + k] * r[i * s.dim() + k];
    mask[i] = (v[i] && std::sqrt(sq) <= thr) ? 1 : 0;
  }
  return CALICO_OK;
}
int32_t oracle_num_effective_parameters(Problem* p, int32_t* n) { oracle::BuildReduced(*p); *n = p->n_eff; return CALICO_OK; }
int32_t oracle_evaluate(Problem* p, double* cost, double* gradient, double* jtj) {
  oracle::BuildReduced(*p);
  std::vector<std::vector<double>> x(p->blocks.size());
  for (size_t i = 0; i < p->blocks.size(); ++i) x[i] = p->blocks[i].v;
  oracle::Evaluation E;
  oracle::Evaluate(*p, x, true, 1, &E);
  if (!E.ok) return p->set_error(CALICO_INTERNAL, "residual evaluation failed");
  if (cost) *cost = E.cost;
  if (gradient) std::copy(E.gradient.begin(), E.gradient.end(), gradient);
  if (jtj) { std::vector<double> H; oracle::AccumulateJtJ(*p, E, nullptr, 1, &H); std::copy(H.begin(), H.end(), jtj); }
  return CALICO_OK;
}
// Dense Jacobian (n_res × n_eff, row-major) and corrected residuals, for
// Jacobian self-verification. apply_loss selects the robustified version.
int32_t oracle_evaluate_jacobian(Problem* p, double* residuals, double* jac_dense) {
  oracle::BuildReduced(*p);
  std::vector<std::vector<double>> x(p->blocks.size());
  for (size_t i = 0; i < p->blocks.size(); ++i) x[i] = p->blocks[i].v;
  oracle::Evaluation E;
  oracle::Evaluate(*p, x, true, 1, &E);
  if (!E.ok) return p->set_error(CALICO_INTERNAL, "residual evaluation failed");
  const int n = p->n_eff;
  if (residuals) std::copy(E.residuals.begin(), E.residuals.end(), residuals);
  if (jac_dense) {
    std::fill(jac_dense, jac_dense + size_t(p->n_res) * n, 0.0);
    for (size_t b = 0; b < p->rblocks.size(); ++b) {
      const oracle::Sensor& s = p->sensors[p->rblocks[b].sensor];
      const double* J = &E.jac[size_t(E.jac_offset[b])];
      int ncols = 0; for (int q = E.col_ptr[b]; q < E.col_ptr[b + 1]; ++q) ncols += E.col_size[q];
      int c0 = 0;
      for (int q = E.col_ptr[b]; q < E.col_ptr[b + 1]; ++q) {
        for (int c = 0; c < E.col_size[q]; ++c)
          for (int k = 0; k < s.dim(); ++k)
            jac_dense[size_t(p->res_offset[b] + k) * n + E.col_off[q] + c] = J[k * ncols + c0 + c];
        c0 += E.col_size[q];
      }
    }
  }
  return CALICO_OK;
}
int64_t oracle_num_residuals(Problem* p) { oracle::BuildReduced(*p); return p->n_res; }
// Cost only (loss applied), as the LM candidate evaluation does.
int32_t oracle_evaluate_cost(Problem* p, int32_t num_threads, double* cost) {
  oracle::BuildReduced(*p);
  std::vector<std::vector<double>> x(p->blocks.size());
  for (size_t i = 0; i < p->blocks.size(); ++i) x[i] = p->blocks[i].v;
  oracle::Evaluation E;
  oracle::Evaluate(*p, x, false, num_threads, &E);
  *cost = E.ok ? E.cost : std::numeric_limits<double>::max();
  return E.ok ? CALICO_OK : CALICO_INTERNAL;
}

// ---- spline + generators (test support) -----------------------------------
typedef oracle::BSpline6 Spline;
Spline* oracle_spline_create() { return new Spline(); }
void oracle_spline_destroy(Spline* s) { delete s; }
// Trajectory::FitSpline (trajectory.cpp:14-49): stamps (any order), quaternions w,x,y,z, translations.
int32_t oracle_spline_fit_poses(Spline* s, int32_t n, const double* stamps, const double* quat_wxyz, const double* trans,
                                double knot_frequency, int32_t order) {
  std::vector<int> idx(n); for (int i = 0; i < n; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return stamps[a] < stamps[b]; });
  std::vector<double> t(n); std::vector<oracle::V3<double>> phi(n);
  for (int i = 0; i < n; ++i) {
    const int j = idx[i]; t[i] = stamps[j];
    phi[i] = oracle::QuaternionToAngleAxisVector(oracle::Quat<double>(quat_wxyz[4 * j], quat_wxyz[4 * j + 1], quat_wxyz[4 * j + 2], quat_wxyz[4 * j + 3]));
  }
  oracle::UnwrapPhaseLogMap(phi);
  std::vector<std::array<double, 6>> data(n);
  for (int i = 0; i < n; ++i) { const int j = idx[i]; data[i] = {phi[i].x, phi[i].y, phi[i].z, trans[3 * j], trans[3 * j + 1], trans[3 * j + 2]}; }
  return s->FitToData(t, data, order, knot_frequency) ? CALICO_OK : CALICO_INVALID_ARGUMENT;
}
// BSpline::FitToData on raw 6-vectors (bspline_test.cpp uses 3 of the 6).
int32_t oracle_spline_fit_vectors(Spline* s, int32_t n, const double* stamps, const double* data6, double knot_frequency,
                                  int32_t order) {
  std::vector<double> t(stamps, stamps + n); std::vector<std::array<double, 6>> d(n);
  for (int i = 0; i < n; ++i) for (int c = 0; c < 6; ++c) d[i][c] = data6[6 * i + c];
  return s->FitToData(t, d, order, knot_frequency) ? CALICO_OK : CALICO_INVALID_ARGUMENT;
}
int32_t oracle_spline_sizes(Spline* s, int32_t* order, int32_t* n_knots, int32_t* n_ctrl, int32_t* n_seg) {
  *order = s->order; *n_knots = int(s->knots.size()); *n_ctrl = int(s->ctrl.size()); *n_seg = int(s->Mi.size()); return CALICO_OK;
}
int32_t oracle_spline_get(Spline* s, double* knots, double* basis, double* ctrl) {
  if (knots) std::copy(s->knots.begin(), s->knots.end(), knots);
  if (basis) for (size_t i = 0; i < s->Mi.size(); ++i) std::copy(s->Mi[i].begin(), s->Mi[i].end(), basis + i * s->order * s->order);
  if (ctrl) for (size_t i = 0; i < s->ctrl.size(); ++i) std::copy(s->ctrl[i].begin(), s->ctrl[i].end(), ctrl + 6 * i);
  return CALICO_OK;
}
int32_t oracle_spline_set_ctrl(Spline* s, const double* ctrl) {
  for (size_t i = 0; i < s->ctrl.size(); ++i) for (int c = 0; c < 6; ++c) s->ctrl[i][c] = ctrl[6 * i + c];
  return CALICO_OK;
}
int32_t oracle_spline_index(Spline* s, double t) { return s->GetSplineIndex(t); }
int32_t oracle_spline_interpolate(Spline* s, int32_t n, const double* times, int32_t derivative, double* out6) {
  for (int i = 0; i < n; ++i) if (!s->Interpolate(times[i], derivative, out6 + 6 * i)) return CALICO_INVALID_ARGUMENT;
  return CALICO_OK;
}
// Camera::Project (camera.cpp:155-208) for one rigid body. Outputs are
// preallocated n_times × n_points; valid=0 where z <= 0 (skipped by the
// reference). out_stamps = t + latency (Q10).
int32_t oracle_project_camera(Spline* s, int32_t model, const double* intr, const double* q_rc_xyzw, const double* t_rc,
                              double latency, int32_t n_times, const double* times, int32_t n_points,
                              const double* points, const double* q_wm_xyzw, const double* t_wm, double* out_pixels,
                              uint8_t* out_valid, double* out_stamps) {
  using namespace oracle;
  const Quat<double> q_rc = Quat<double>::FromCoeffs(q_rc_xyzw), q_wm = Quat<double>::FromCoeffs(q_wm_xyzw);
  const V3<double> trc(t_rc[0], t_rc[1], t_rc[2]), twm(t_wm[0], t_wm[1], t_wm[2]);
  for (int i = 0; i < n_times; ++i) {
    double pv[6];
    if (!s->Interpolate(times[i], 0, pv)) return CALICO_INVALID_ARGUMENT;
    // Trajectory::VectorToPose3 (trajectory.h:93-101)
    const Quat<double> q_wr = AngleAxisToQuaternion(V3<double>(pv[0], pv[1], pv[2]));
    const V3<double> t_wr(pv[3], pv[4], pv[5]);
    // T_world_cam = T_world_rig * T_rig_cam (typedefs.h:99-108); inverse typedefs.h:125-129
    const Quat<double> q_wc = q_wr * q_rc;
    const V3<double> t_wc = q_wr * trc + t_wr;
    const Quat<double> q_cw = q_wc.conjugate();
    const V3<double> t_cw = -(q_cw * t_wc);
    // T_camera_rigidbody = T_camera_world * T_world_rigidbody
    const Quat<double> q_cb = q_cw * q_wm;
    const V3<double> t_cb = q_cw * twm + t_cw;
    for (int j = 0; j < n_points; ++j) {
      const V3<double> pt(points[3 * j], points[3 * j + 1], points[3 * j + 2]);
      const V3<double> pc = q_cb * pt + t_cb;
      const size_t o = size_t(i) * n_points + j;
      out_stamps[o] = times[i] + latency;
      if (pc.z <= 0) { out_valid[o] = 0; out_pixels[2 * o] = out_pixels[2 * o + 1] = 0; continue; }
      double px[2] = {0, 0};
      const bool ok = ProjectPoint<double>(model, intr, pc, px);
      out_valid[o] = ok ? 1 : 0;  // the reference dereferences the StatusOr unchecked
      out_pixels[2 * o] = px[0]; out_pixels[2 * o + 1] = px[1];
    }
  }
  return CALICO_OK;
}
// Gyroscope::Project (gyroscope.cpp:56-82)
int32_t oracle_project_gyroscope(Spline* s, int32_t model, const double* intr, const double* q_rg_xyzw, double latency,
                                 int32_t n, const double* times, double* out_meas, double* out_stamps) {
  using namespace oracle;
  const Quat<double> q_rg = Quat<double>::FromCoeffs(q_rg_xyzw);
  for (int i = 0; i < n; ++i) {
    double p[6], pd[6];
    if (!s->Interpolate(times[i], 0, p) || !s->Interpolate(times[i], 1, pd)) return CALICO_INVALID_ARGUMENT;
    const V3<double> phi(-p[0], -p[1], -p[2]), phid(-pd[0], -pd[1], -pd[2]);
    const M3<double> J = ExpSO3Jacobian(phi);
    const V3<double> om = J * phid;
    const V3<double> og = -(q_rg.inverse() * om);
    V3<double> pr;
    if (!ImuProject<double>(model, intr, og, &pr)) return CALICO_INVALID_ARGUMENT;
    out_meas[3 * i] = pr.x; out_meas[3 * i + 1] = pr.y; out_meas[3 * i + 2] = pr.z;
    out_stamps[i] = times[i] + latency;
  }
  return CALICO_OK;
}
// Accelerometer::Project (accelerometer.cpp:76-123)
int32_t oracle_project_accelerometer(Spline* s, int32_t model, const double* intr, const double* q_ra_xyzw,
                                     const double* t_ra, double latency, const double* gravity, int32_t n,
                                     const double* times, double* out_meas, double* out_stamps) {
  using namespace oracle;
  const Quat<double> q_ra = Quat<double>::FromCoeffs(q_ra_xyzw);
  const V3<double> tra(t_ra[0], t_ra[1], t_ra[2]), g(gravity[0], gravity[1], gravity[2]);
  for (int i = 0; i < n; ++i) {
    double p[6], pd[6], pdd[6];
    if (!s->Interpolate(times[i], 0, p) || !s->Interpolate(times[i], 1, pd) || !s->Interpolate(times[i], 2, pdd))
      return CALICO_INVALID_ARGUMENT;
    const V3<double> phi(-p[0], -p[1], -p[2]), phid(-pd[0], -pd[1], -pd[2]), phidd(-pdd[0], -pdd[1], -pdd[2]);
    const Quat<double> q_rw = AngleAxisToQuaternion(phi);
    const V3<double> ddt(pdd[3], pdd[4], pdd[5]);
    const M3<double> J = ExpSO3Jacobian(phi);
    const M3<double> Jdot = ExpSO3JacobianDot(phi, phid);
    const V3<double> om = J * phid;
    const V3<double> al = Jdot * phid + J * phidd;
    const M3<double> Alpha = -Skew(al), Omega = -Skew(om);
    const V3<double> f = q_ra.inverse() * (q_rw * (ddt - g) + (Omega * Omega + Alpha) * tra);
    V3<double> pr;
    if (!ImuProject<double>(model, intr, f, &pr)) return CALICO_INVALID_ARGUMENT;
    out_meas[3 * i] = pr.x; out_meas[3 * i + 1] = pr.y; out_meas[3 * i + 2] = pr.z;
    out_stamps[i] = times[i] + latency;
  }
  return CALICO_OK;
}

// ---- geometry / model known-answer entry points (unit tests) ---------------
void oracle_exp_so3(const double* phi, double* R9) {
  const oracle::M3<double> R = oracle::ExpSO3(oracle::V3<double>(phi[0], phi[1], phi[2]));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R9[3 * i + j] = R.m[i][j];
}
void oracle_ln_so3(const double* R9, double* phi) {
  oracle::M3<double> R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R.m[i][j] = R9[3 * i + j];
  const oracle::V3<double> p = oracle::LnSO3(R); phi[0] = p.x; phi[1] = p.y; phi[2] = p.z;
}
void oracle_exp_so3_jacobian(const double* phi, double* J9) {
  const oracle::M3<double> J = oracle::ExpSO3Jacobian(oracle::V3<double>(phi[0], phi[1], phi[2]));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J9[3 * i + j] = J.m[i][j];
}
void oracle_exp_so3_jacobian_dot(const double* phi, const double* phid, double* J9) {
  const oracle::M3<double> J = oracle::ExpSO3JacobianDot(oracle::V3<double>(phi[0], phi[1], phi[2]), oracle::V3<double>(phid[0], phid[1], phid[2]));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J9[3 * i + j] = J.m[i][j];
}
void oracle_exp_so3_hessian(const double* phi, double* H27) {
  oracle::M3<double> H[3]; oracle::ExpSO3Hessian(oracle::V3<double>(phi[0], phi[1], phi[2]), H);
  for (int a = 0; a < 3; ++a) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H27[9 * a + 3 * i + j] = H[a].m[i][j];
}
int32_t oracle_project_point(int32_t model, const double* intr, const double* pt, double* px) {
  return oracle::ProjectPoint<double>(model, intr, oracle::V3<double>(pt[0], pt[1], pt[2]), px) ? CALICO_OK : CALICO_INVALID_ARGUMENT;
}
int32_t oracle_imu_project(int32_t model, const double* intr, const double* w, double* out) {
  oracle::V3<double> o;
  if (!oracle::ImuProject<double>(model, intr, oracle::V3<double>(w[0], w[1], w[2]), &o)) return CALICO_INVALID_ARGUMENT;
  out[0] = o.x; out[1] = o.y; out[2] = o.z; return CALICO_OK;
}
int32_t oracle_camera_num_params(int32_t model) { return oracle::CameraNumParams(model); }
int32_t oracle_imu_num_params(int32_t model) { return oracle::ImuNumParams(model); }
void oracle_angle_axis_to_quaternion(const double* aa, double* q_wxyz) {
  const oracle::Quat<double> q = oracle::AngleAxisToQuaternion(oracle::V3<double>(aa[0], aa[1], aa[2]));
  q_wxyz[0] = q.w; q_wxyz[1] = q.x; q_wxyz[2] = q.y; q_wxyz[3] = q.z;
}

}  // extern "C"
