// Restatement of the reference's tests against the C++ facade (include/calico/calico.hpp):
//   calico/test/batch_optimizer_test.cpp:32-213  ToyStereoCameraAndImuCalibration (needs a GPU)
//   calico/test/trajectory_test.cpp:23-34        fit + interpolate reproduces the poses within 1e-3
//   calico/test/camera_test.cpp:82-101           duplicate measurements -> kInvalidArgument, unique ones added
//   calico/test/typedefs_test.cpp:37-60          accessor pointer stability
// Usage: toy_stereo_imu [--host-only]
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

#include "calico/calico.hpp"

using namespace calico;

// --host-only runs on machines without a GPU: the spline fit is then done by the CPU oracle (oracle/ is the tests'
// checker), installed through BSpline6::fit_solver(). The GPU run keeps the default, calico_fit_spline.
extern "C" {
void* oracle_spline_create();
void oracle_spline_destroy(void*);
int32_t oracle_spline_fit_vectors(void*, int32_t n, const double* stamps, const double* data6, double knot_frequency, int32_t order);
int32_t oracle_spline_sizes(void*, int32_t* order, int32_t* n_knots, int32_t* n_ctrl, int32_t* n_seg);
int32_t oracle_spline_get(void*, double* knots, double* basis, double* ctrl);
}
static int32_t oracle_fit(int32_t order, int32_t n_knots, const double* knots, const double*, int64_t n, const double* stamps,
                          const double* data6, double* ctrl_out) {
  const double kf = std::round(1e6 / (knots[order] - knots[order - 1])) * 1e-6;
  void* s = oracle_spline_create();
  int32_t rc = oracle_spline_fit_vectors(s, int32_t(n), stamps, data6, kf, order), o = 0, nk = 0, nc = 0, ns = 0;
  if (rc == 0) rc = oracle_spline_sizes(s, &o, &nk, &nc, &ns);
  if (rc == 0 && (nk != n_knots || o != order)) rc = CALICO_INTERNAL;
  if (rc == 0) {
    std::vector<double> k, b; k.resize(size_t(nk)); b.resize(size_t(ns) * o * o);
    rc = oracle_spline_get(s, k.data(), b.data(), ctrl_out);
  }
  oracle_spline_destroy(s);
  return rc == 0 ? CALICO_OK : CALICO_INTERNAL;
}

static int failures = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) { std::printf("CHECK FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

// test_utils.h:11-116 DefaultSyntheticTest
struct DefaultSyntheticTest {
  std::map<double, Pose3d> trajectory;
  std::vector<double> stamps;
  std::vector<Vector3d> points;
  DefaultSyntheticTest() {
    const double kDeg2Rad = M_PI / 180.0;
    const Quaterniond q0 = Quaterniond::FromAngleAxis(M_PI, Vector3d(0, 0, 1)) * Quaterniond::FromAngleAxis(M_PI, Vector3d(1, 0, 0));
    const Vector3d t0(0, 0, 1);
    const double ang[5] = {0, 30 * kDeg2Rad, 0, -30 * kDeg2Rad, 0}, pos[5] = {0, 0.5, 0, -0.5, 0};
    const int n = 10;
    const double dti = 1.0 / n, dta = dti * 0.75;
    double interp[10];
    for (int i = 0; i < n; ++i) interp[i] = (std::sin(dti * i * M_PI - M_PI_2) + 1.0) / 2.0;
    double t = 0;
    for (int ax = 0; ax < 3; ++ax) {
      Vector3d axis(ax == 0, ax == 1, ax == 2);
      for (int i = 1; i < 5; ++i) for (int s = 0; s < n; ++s) {
        const double th = (ang[i] - ang[i - 1]) * interp[s] + ang[i - 1];
        trajectory[t] = Pose3d(q0 * Quaterniond::FromAngleAxis(th, axis), t0); t += dta;
      }
      for (int i = 1; i < 5; ++i) for (int s = 0; s < n; ++s) {
        const double p = (pos[i] - pos[i - 1]) * interp[s] + pos[i - 1];
        trajectory[t] = Pose3d(q0, p * axis + t0); t += dta;
      }
    }
    for (const auto& kv : trajectory) stamps.push_back(kv.first);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) points.push_back(Vector3d(i * 0.3 - 0.75, j * 0.3 - 0.75, 0.0));
  }
};

static double maxdiff(const VectorXd& a, const VectorXd& b) { double m = 0; for (size_t i = 0; i < a.size(); ++i) m = std::max(m, std::fabs(a[i] - b[i])); return m; }
static double posediff(const Pose3d& a, const Pose3d& b) {
  double m = 0;
  for (int i = 0; i < 4; ++i) m = std::max(m, std::fabs(a.rotation().data()[i] - b.rotation().data()[i]));
  for (int i = 0; i < 3; ++i) m = std::max(m, std::fabs(a.translation()[i] - b.translation()[i]));
  return m;
}

int main(int argc, char** argv) {
  const bool host_only = argc > 1 && !std::strcmp(argv[1], "--host-only");
  if (host_only) BSpline6::fit_solver() = &oracle_fit;
  DefaultSyntheticTest fixture;
  // typedefs_test.cpp:37-60
  { Pose3d p; double* a = p.rotation().coeffs().data(); double* b = p.translation().data(); Pose3d q = p; (void)q;
    CHECK(a == p.rotation().coeffs().data() && b == p.translation().data()); }
  // trajectory_test.cpp:23-34
  Trajectory* trajectory = new Trajectory;
  CHECK(trajectory->FitSpline(fixture.trajectory).ok());
  CHECK(trajectory->spline().control_points().size() == 185);  // 181 valid knots, order 6
  { auto poses = trajectory->Interpolate(fixture.stamps); CHECK(poses.ok());
    double worst = 0; size_t i = 0;
    for (const auto& kv : fixture.trajectory) {
      const Pose3d& e = kv.second; const Pose3d& a = (*poses)[i++];
      // quaternion sign is not unique
      double d = std::min(posediff(e, a), posediff(Pose3d(Quaterniond(-e.rotation().w(), -e.rotation().x(), -e.rotation().y(), -e.rotation().z()), e.translation()), a));
      worst = std::max(worst, d);
    }
    CHECK(worst < 1e-3); }
  CHECK(!trajectory->Interpolate({-1.0}).ok() && trajectory->Interpolate({-1.0}).status().code() == StatusCode::kInvalidArgument);

  RigidBody planar_target; planar_target.world_pose_is_constant = true; planar_target.model_definition_is_constant = true;
  for (size_t i = 0; i < fixture.points.size(); ++i) planar_target.model_definition[int(i)] = fixture.points[i];
  WorldModel* world_model = new WorldModel;
  const Vector3d true_gravity = world_model->gravity();
  CHECK(world_model->AddRigidBody(&planar_target, /*take_ownership=*/false).ok());
  CHECK(world_model->AddRigidBody(&planar_target, false).code() == StatusCode::kInvalidArgument);  // world_model_test.cpp

  // ground truth (fixed "random" draws; the reference uses unseeded Eigen::Random)
  const VectorXd true_cam = {785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4, -1.853e-2};
  const double ang = 2.0 * M_PI / 180.0;
  auto unit = [](double a, double b, double c) { const double n = std::sqrt(a * a + b * b + c * c); return Vector3d(a / n, b / n, c / n); };
  Pose3d ex_left, ex_right(Quaterniond::FromAngleAxis(ang, unit(0.68, -0.21, 0.57)), 0.05 * Vector3d(0.6, -0.33, 0.54));
  Pose3d ex_gyro(Quaterniond::FromAngleAxis(ang, unit(-0.44, 0.11, -0.05)), Vector3d());
  Pose3d ex_acc(Quaterniond::FromAngleAxis(ang, unit(0.26, -0.27, 0.9)), Vector3d());
  const VectorXd true_imu = {1.3, 0.01, -0.01, 0.01};

  sensors::Camera true_left, true_right;
  CHECK(true_left.SetIntrinsics(true_cam).code() == StatusCode::kInvalidArgument);  // model not set (camera.cpp:24-27)
  CHECK(true_left.SetModel(sensors::CameraIntrinsicsModel::kOpenCv5).ok() && true_left.SetIntrinsics(true_cam).ok());
  CHECK(true_left.SetIntrinsics({1, 2, 3}).code() == StatusCode::kInvalidArgument);
  true_left.SetExtrinsics(ex_left);
  CHECK(true_right.SetModel(sensors::CameraIntrinsicsModel::kOpenCv5).ok() && true_right.SetIntrinsics(true_cam).ok());
  true_right.SetExtrinsics(ex_right); CHECK(true_right.SetLatency(0.01).ok());
  auto m_left = true_left.Project(fixture.stamps, *trajectory, *world_model);
  auto m_right = true_right.Project(fixture.stamps, *trajectory, *world_model);
  CHECK(m_left.ok() && m_right.ok());
  CHECK(m_left->size() == 240 * 36);  // camera_test.cpp visibility: all points in front of the camera
  sensors::Gyroscope true_gyro; sensors::Accelerometer true_acc;
  CHECK(true_gyro.SetModel(sensors::GyroscopeIntrinsicsModel::kGyroscopeScaleAndBias).ok() && true_gyro.SetIntrinsics(true_imu).ok());
  true_gyro.SetExtrinsics(ex_gyro); true_gyro.SetLatency(0.02);
  CHECK(true_acc.SetModel(sensors::AccelerometerIntrinsicsModel::kAccelerometerScaleAndBias).ok() && true_acc.SetIntrinsics(true_imu).ok());
  true_acc.SetExtrinsics(ex_acc); true_acc.SetLatency(0.02);
  auto m_gyro = true_gyro.Project(fixture.stamps, *trajectory, *world_model);
  auto m_acc = true_acc.Project(fixture.stamps, *trajectory, *world_model);
  CHECK(m_gyro.ok() && m_acc.ok());

  // sensors to optimise (batch_optimizer_test.cpp:124-172)
  VectorXd init_cam = true_cam; for (double& v : init_cam) v *= 1.01; for (int i = 3; i < 8; ++i) init_cam[size_t(i)] = 0.0;
  Pose3d init_right = ex_right; init_right.translation() = init_right.translation() + 0.01 * Vector3d(0.3, -0.8, 0.5);
  auto* camera_left = new sensors::Camera; camera_left->SetName("Left");
  CHECK(camera_left->SetModel(sensors::CameraIntrinsicsModel::kOpenCv5).ok() && camera_left->SetIntrinsics(init_cam).ok());
  camera_left->EnableExtrinsicsEstimation(false); camera_left->EnableIntrinsicsEstimation(true); camera_left->EnableLatencyEstimation(false);
  CHECK(camera_left->AddMeasurements(*m_left).ok());
  CHECK(camera_left->AddMeasurements({(*m_left)[0]}).code() == StatusCode::kInvalidArgument);  // camera_test.cpp:82-101
  CHECK(camera_left->NumberOfMeasurements() == int(m_left->size()));
  CHECK(camera_left->SetMeasurementNoise(0.0).code() == StatusCode::kInvalidArgument);  // camera.cpp:62-68
  auto* camera_right = new sensors::Camera; camera_right->SetName("Right");
  CHECK(camera_right->SetModel(sensors::CameraIntrinsicsModel::kOpenCv5).ok() && camera_right->SetIntrinsics(init_cam).ok());
  camera_right->SetExtrinsics(init_right);
  camera_right->EnableExtrinsicsEstimation(true); camera_right->EnableIntrinsicsEstimation(true); camera_right->EnableLatencyEstimation(true);
  CHECK(camera_right->AddMeasurements(*m_right).ok());
  VectorXd init_imu = true_imu; for (double& v : init_imu) v *= 1.01;
  auto* gyro = new sensors::Gyroscope; gyro->SetName("Gyroscope");
  CHECK(gyro->SetModel(sensors::GyroscopeIntrinsicsModel::kGyroscopeScaleAndBias).ok() && gyro->SetIntrinsics(init_imu).ok());
  gyro->SetExtrinsics(ex_gyro);
  gyro->EnableExtrinsicsEstimation(true); gyro->EnableIntrinsicsEstimation(true); gyro->EnableLatencyEstimation(true);
  CHECK(gyro->AddMeasurements(*m_gyro).ok());
  Pose3d init_acc = ex_acc; init_acc.translation() = init_acc.translation() + 0.05 * Vector3d(-0.2, 0.7, 0.4);
  auto* acc = new sensors::Accelerometer; acc->SetName("Accelerometer");
  CHECK(acc->SetModel(sensors::AccelerometerIntrinsicsModel::kAccelerometerScaleAndBias).ok() && acc->SetIntrinsics(init_imu).ok());
  acc->SetExtrinsics(init_acc);
  acc->EnableExtrinsicsEstimation(true); acc->EnableIntrinsicsEstimation(true); acc->EnableLatencyEstimation(true);
  CHECK(acc->AddMeasurements(*m_acc).ok());

  // Problem bookkeeping without solving (world_model_test.cpp / camera_test.cpp NumParameters)
  { Problem problem;
    CHECK(world_model->AddParametersToProblem(problem) == 36 * 3 + 7 + 3);
    CHECK(trajectory->AddParametersToProblem(problem) == 185 * 6);
    auto np = camera_left->AddParametersToProblem(problem); CHECK(np.ok() && *np == 8 + 7 + 1);
    auto nr = camera_left->AddResidualsToProblem(problem, *trajectory, *world_model); CHECK(nr.ok() && *nr == int(m_left->size()));
    sensors::Camera no_model; CHECK(no_model.AddParametersToProblem(problem).status().code() == StatusCode::kFailedPrecondition); }

  if (!host_only) {
    BatchOptimizer optimizer;
    optimizer.AddSensor(camera_left); optimizer.AddSensor(camera_right); optimizer.AddSensor(gyro); optimizer.AddSensor(acc);
    optimizer.AddWorldModel(world_model); optimizer.AddTrajectory(trajectory);
    SolverOptions options = DefaultSolverOptions();
    options.minimizer_progress_to_stdout = 0;
    options.max_num_iterations = 100;  // the toy problem is sensitive to the random draw; the reference's default is 50
    auto summary = optimizer.Optimize(options);
    CHECK(summary.ok());
    if (summary.ok()) {
      const double kSmallNumber = 1e-7;
      std::printf("%s\n", summary->FullReport().c_str());
      CHECK(summary->termination_type == CALICO_CONVERGENCE);
      CHECK(summary->final_cost < kSmallNumber);
      CHECK(maxdiff(true_cam, camera_left->GetIntrinsics()) < kSmallNumber);
      CHECK(maxdiff(true_cam, camera_right->GetIntrinsics()) < kSmallNumber);
      CHECK(posediff(ex_right, camera_right->GetExtrinsics()) < kSmallNumber);
      CHECK(std::fabs(0.01 - camera_right->GetLatency()) < kSmallNumber);
      CHECK(maxdiff(true_imu, gyro->GetIntrinsics()) < kSmallNumber);
      CHECK(posediff(ex_gyro, gyro->GetExtrinsics()) < kSmallNumber);
      CHECK(std::fabs(0.02 - gyro->GetLatency()) < kSmallNumber);
      CHECK(maxdiff(true_imu, acc->GetIntrinsics()) < kSmallNumber);
      CHECK(posediff(ex_acc, acc->GetExtrinsics()) < kSmallNumber);
      CHECK(std::fabs(0.02 - acc->GetLatency()) < kSmallNumber);
      CHECK(std::fabs(world_model->gravity()[2] - true_gravity[2]) < kSmallNumber);
      auto pairs = camera_left->GetMeasurementResidualPairs();
      CHECK(pairs.ok() && pairs->size() == m_left->size());
      double worst = 0; for (const auto& pr : *pairs) worst = std::max(worst, std::max(std::fabs(pr.second.x()), std::fabs(pr.second.y())));
      CHECK(worst < 1e-5);
      std::printf("final cost %.3e, %d iterations, max |camera residual| %.2e\n", summary->final_cost, summary->num_iterations, worst);
    }
    // BatchOptimizer owns sensors / world model / trajectory (take_ownership = true)
  } else {
    delete camera_left; delete camera_right; delete gyro; delete acc; delete world_model; delete trajectory;
  }
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}
