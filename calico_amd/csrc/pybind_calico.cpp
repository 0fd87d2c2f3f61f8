// pybind_calico.cpp — the Python face of the host-side mirror (include/calico/calico.hpp): the class, method and
// enum names of the reference's `calico` module (calico/calico.cpp:18-437) over the HIP backend, so that the
// notebooks' call sequence (build sensors -> FitSpline -> BatchOptimizer.Optimize -> residual pairs -> outliers)
// runs unchanged with `from calico_amd import calico`. Conventions taken from the reference binding:
//   * Status-returning setters that the reference wraps in lambdas raise RuntimeError("Error: <message>"),
//     the others return the Status object (SetModel, SetLatency, AddMeasurement(s), SetMeasurementNoise),
//   * Pose3d.rotation is [w, x, y, z], vectors travel as numpy arrays,
//   * AddSensor / AddTrajectory / AddWorldModel / AddRigidBody / AddLandmark do not take ownership.
// Not bound: AprilGridDetector (OpenCV image processing, outside the optimisation path).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <memory>
#include <sstream>
#include <stdexcept>

#include "calico/calico.hpp"

namespace py = pybind11;

namespace pybind11 {
namespace detail {
template <int N, class V>
struct fixed_vector_caster {
  static bool load_into(handle src, V& value) {
    auto arr = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(src);
    if (!arr || arr.size() != N) return false;
    for (int i = 0; i < N; ++i) value.v[i] = arr.data()[i];
    return true;
  }
  static handle to_python(const V& v) {
    py::array_t<double> a(N);
    for (int i = 0; i < N; ++i) a.mutable_data()[i] = v.v[i];
    return a.release();
  }
};
template <>
struct type_caster<calico::Vector3d> {
  PYBIND11_TYPE_CASTER(calico::Vector3d, const_name("numpy.ndarray[float64[3]]"));
  bool load(handle src, bool) { return fixed_vector_caster<3, calico::Vector3d>::load_into(src, value); }
  static handle cast(const calico::Vector3d& v, return_value_policy, handle) { return fixed_vector_caster<3, calico::Vector3d>::to_python(v); }
};
template <>
struct type_caster<calico::Vector2d> {
  PYBIND11_TYPE_CASTER(calico::Vector2d, const_name("numpy.ndarray[float64[2]]"));
  bool load(handle src, bool) { return fixed_vector_caster<2, calico::Vector2d>::load_into(src, value); }
  static handle cast(const calico::Vector2d& v, return_value_policy, handle) { return fixed_vector_caster<2, calico::Vector2d>::to_python(v); }
};
}  // namespace detail
}  // namespace pybind11

namespace {

using namespace calico;           // NOLINT
using namespace calico::sensors;  // NOLINT

void raise_if_error(const Status& st) {
  if (!st.ok()) throw std::runtime_error(std::string("Error: ") + st.message());
}
py::array_t<double> to_array(const VectorXd& v) {
  py::array_t<double> a(py::ssize_t(v.size()));
  std::copy(v.begin(), v.end(), a.mutable_data());
  return a;
}
VectorXd to_vector(const py::object& o) {
  auto arr = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(o);
  if (!arr) throw py::type_error("expected a sequence of floats");
  return VectorXd(arr.data(), arr.data() + arr.size());
}

// the part of the sensor interface the three sensors share (calico.cpp:77-103 and its two repetitions)
template <class S, class PyClass>
void bind_sensor_common(PyClass& c) {
  c.def(py::init<>())
      .def("SetName", &S::SetName)
      .def("GetName", &S::GetName)
      .def("SetExtrinsics", &S::SetExtrinsics)
      .def("GetExtrinsics", &S::GetExtrinsics)
      .def("SetIntrinsics", [](S& self, const py::object& intrinsics) { raise_if_error(self.SetIntrinsics(to_vector(intrinsics))); })
      .def("GetIntrinsics", [](const S& self) { return to_array(self.GetIntrinsics()); })
      .def("SetLatency", &S::SetLatency)
      .def("GetLatency", &S::GetLatency)
      .def("EnableExtrinsicsEstimation", &S::EnableExtrinsicsEstimation)
      .def("EnableIntrinsicsEstimation", &S::EnableIntrinsicsEstimation)
      .def("EnableLatencyEstimation", &S::EnableLatencyEstimation)
      .def("SetModel", &S::SetModel)
      .def("GetModel", &S::GetModel)
      .def("SetLossFunction", &S::SetLossFunction, py::arg("loss"), py::arg("scale") = 1.0)
      .def("AddMeasurement", &S::AddMeasurement)
      .def("AddMeasurements", &S::AddMeasurements)
      .def("SetMeasurementNoise", &S::SetMeasurementNoise)
      .def("NumberOfMeasurements", &S::NumberOfMeasurements)
      .def("ClearMeasurements", &S::ClearMeasurements)
      .def("Project",
           [](const S& self, const std::vector<double>& interp_times, const Trajectory& sensorrig_trajectory,
              const WorldModel& world_model) {
             auto vals = self.Project(interp_times, sensorrig_trajectory, world_model);
             raise_if_error(vals.status());
             return vals.value();
           });
}

}  // namespace

PYBIND11_MODULE(_calico, m) {
  m.doc() = "calico_amd: the reference's `calico` Python API over the MI355X HIP backend (libcalico_hip.so)";

  py::enum_<StatusCode>(m, "StatusCode")
      .value("kOk", StatusCode::kOk)
      .value("kInvalidArgument", StatusCode::kInvalidArgument)
      .value("kFailedPrecondition", StatusCode::kFailedPrecondition)
      .value("kUnimplemented", StatusCode::kUnimplemented)
      .value("kInternal", StatusCode::kInternal);

  py::class_<Status>(m, "Status")
      .def(py::init<>())
      .def("ok", &Status::ok)
      .def("code", &Status::code)
      .def("message", [](const Status& self) { return std::string(self.message()); });

  py::class_<Pose3d>(m, "Pose3d")
      .def(py::init<>())
      .def(py::init<const Pose3d&>())
      .def_property(
          "rotation", [](const Pose3d& self) { const auto q = self.GetRotation(); return to_array(VectorXd(q.begin(), q.end())); },
          [](Pose3d& self, const py::object& wxyz) {
            const VectorXd q = to_vector(wxyz);
            if (q.size() != 4) throw py::value_error("rotation is [w, x, y, z]");
            self.SetRotation({q[0], q[1], q[2], q[3]});
          })
      .def_property("translation", &Pose3d::GetTranslation, &Pose3d::SetTranslation)
      .def("__copy__", [](const Pose3d& self) { return Pose3d(self); })
      .def("__deepcopy__", [](const Pose3d& self, const py::dict&) { return Pose3d(self); });

  py::enum_<utils::LossFunctionType>(m, "LossFunctionType")
      .value("kNone", utils::LossFunctionType::kNone)
      .value("kHuber", utils::LossFunctionType::kHuber)
      .value("kCauchy", utils::LossFunctionType::kCauchy);

  py::class_<Sensor, std::shared_ptr<Sensor>>(m, "Sensor");  // NOLINT

  // ---- IMU: the gyroscope and accelerometer id / measurement types are one C++ type, bound once ----
  py::class_<ImuObservationId>(m, "GyroscopeObservationId")
      .def(py::init<>())
      .def(py::init<const ImuObservationId&>())
      .def_readwrite("stamp", &ImuObservationId::stamp)
      .def_readwrite("sequence", &ImuObservationId::sequence)
      .def("__eq__", [](const ImuObservationId& a, const ImuObservationId& b) { return a == b; })
      .def("__hash__", [](const ImuObservationId& a) { return ImuObservationIdHash()(a); });
  py::class_<ImuMeasurement>(m, "GyroscopeMeasurement")
      .def(py::init<>())
      .def(py::init<const ImuMeasurement&>())
      .def_readwrite("measurement", &ImuMeasurement::measurement)
      .def_readwrite("id", &ImuMeasurement::id);
  m.attr("AccelerometerObservationId") = m.attr("GyroscopeObservationId");
  m.attr("AccelerometerMeasurement") = m.attr("GyroscopeMeasurement");

  py::enum_<AccelerometerIntrinsicsModel>(m, "AccelerometerIntrinsicsModel")
      .value("kNone", AccelerometerIntrinsicsModel::kNone)
      .value("kAccelerometerScaleOnly", AccelerometerIntrinsicsModel::kAccelerometerScaleOnly)
      .value("kAccelerometerScaleAndBias", AccelerometerIntrinsicsModel::kAccelerometerScaleAndBias)
      .value("kAccelerometerVectorNav", AccelerometerIntrinsicsModel::kAccelerometerVectorNav);
  py::class_<Accelerometer, std::shared_ptr<Accelerometer>, Sensor> accelerometer(m, "Accelerometer");
  bind_sensor_common<Accelerometer>(accelerometer);

  py::enum_<GyroscopeIntrinsicsModel>(m, "GyroscopeIntrinsicsModel")
      .value("kNone", GyroscopeIntrinsicsModel::kNone)
      .value("kGyroscopeScaleOnly", GyroscopeIntrinsicsModel::kGyroscopeScaleOnly)
      .value("kGyroscopeScaleAndBias", GyroscopeIntrinsicsModel::kGyroscopeScaleAndBias)
      .value("kGyroscopeVectorNav", GyroscopeIntrinsicsModel::kGyroscopeVectorNav);
  py::class_<Gyroscope, std::shared_ptr<Gyroscope>, Sensor> gyroscope(m, "Gyroscope");
  bind_sensor_common<Gyroscope>(gyroscope);

  // ---- camera ----
  py::enum_<CameraIntrinsicsModel>(m, "CameraIntrinsicsModel")
      .value("kNone", CameraIntrinsicsModel::kNone)
      .value("kOpenCv5", CameraIntrinsicsModel::kOpenCv5)
      .value("kOpenCv8", CameraIntrinsicsModel::kOpenCv8)
      .value("kKannalaBrandt", CameraIntrinsicsModel::kKannalaBrandt)
      .value("kDoubleSphere", CameraIntrinsicsModel::kDoubleSphere)
      .value("kFieldOfView", CameraIntrinsicsModel::kFieldOfView)
      .value("kUnifiedCamera", CameraIntrinsicsModel::kUnifiedCamera)
      .value("kExtendedUnifiedCamera", CameraIntrinsicsModel::kExtendedUnifiedCamera);

  py::class_<CameraObservationId>(m, "CameraObservationId")
      .def(py::init([] { return CameraObservationId{0.0, 0, 0, 0}; }))
      .def(py::init<const CameraObservationId&>())
      .def("__hash__", [](const CameraObservationId& id) { return CameraObservationIdHash()(id); })
      .def("__eq__", [](const CameraObservationId& a, const CameraObservationId& b) { return a == b; })
      .def("__str__",
           [](const CameraObservationId& id) {
             std::ostringstream os;
             os << "stamp: " << id.stamp << ", image_id: " << id.image_id << ", model_id: " << id.model_id
                << ", feature_id: " << id.feature_id;
             return os.str();
           })
      .def_readwrite("stamp", &CameraObservationId::stamp)
      .def_readwrite("image_id", &CameraObservationId::image_id)
      .def_readwrite("model_id", &CameraObservationId::model_id)
      .def_readwrite("feature_id", &CameraObservationId::feature_id);

  py::class_<CameraMeasurement>(m, "CameraMeasurement")
      .def(py::init([] { return CameraMeasurement{Vector2d(), CameraObservationId{0.0, 0, 0, 0}}; }))
      .def(py::init<const CameraMeasurement&>())
      .def_readwrite("pixel", &CameraMeasurement::pixel)
      .def_readwrite("id", &CameraMeasurement::id);

  py::class_<Camera, std::shared_ptr<Camera>, Sensor> camera(m, "Camera");
  bind_sensor_common<Camera>(camera);
  camera
      .def("GetMeasurementResidualPairs",
           [](const Camera& self) {
             auto pairs = self.GetMeasurementResidualPairs();
             raise_if_error(pairs.status());
             return pairs.value();
           })
      .def("GetMeasurementIdToMeasurement",
           [](const Camera& self) {
             py::dict out;
             for (const auto& kv : self.GetMeasurementIdToMeasurement()) out[py::cast(kv.first)] = py::cast(kv.second);
             return out;
           })
      .def("MarkOutlierById", [](Camera& self, const CameraObservationId& id) { raise_if_error(self.MarkOutlierById(id)); })
      .def("MarkOutliersById",
           [](Camera& self, const std::vector<CameraObservationId>& ids) { raise_if_error(self.MarkOutliersById(ids)); })
      .def("ClearOutliersList", &Camera::ClearOutliersList);

  // ---- trajectory, world model ----
  py::class_<Trajectory, std::shared_ptr<Trajectory>>(m, "Trajectory")
      .def(py::init<>())
      .def(
          "FitSpline",
          [](Trajectory& self, const std::map<double, Pose3d>& poses, double knot_frequency, int spline_order) {
            raise_if_error(self.FitSpline(poses, knot_frequency, spline_order));
          },
          py::arg("poses"), py::arg("knot_frequency") = Trajectory::kDefaultKnotFrequency,
          py::arg("spline_order") = Trajectory::kDefaultSplineOrder)
      .def("Interpolate", [](const Trajectory& self, const std::vector<double>& stamps) {
        auto poses = self.Interpolate(stamps);
        raise_if_error(poses.status());
        return poses.value();
      });

  py::class_<Landmark, std::shared_ptr<Landmark>>(m, "Landmark")
      .def(py::init<>())
      .def_readwrite("point", &Landmark::point)
      .def_readwrite("id", &Landmark::id)
      .def_readwrite("point_is_constant", &Landmark::point_is_constant);

  py::class_<RigidBody, std::shared_ptr<RigidBody>>(m, "RigidBody")
      .def(py::init<>())
      .def_readwrite("model_definition", &RigidBody::model_definition)
      .def_readwrite("T_world_rigidbody", &RigidBody::T_world_rigidbody)
      .def_readwrite("id", &RigidBody::id)
      .def_readwrite("world_pose_is_constant", &RigidBody::world_pose_is_constant)
      .def_readwrite("model_definition_is_constant", &RigidBody::model_definition_is_constant);

  py::class_<WorldModel, std::shared_ptr<WorldModel>>(m, "WorldModel")
      .def(py::init<>())
      .def("AddLandmark",
           [](WorldModel& self, std::shared_ptr<Landmark> landmark) {
             raise_if_error(self.AddLandmark(landmark.get(), /*take_ownership=*/false));
           },
           py::keep_alive<1, 2>())
      .def("AddRigidBody",
           [](WorldModel& self, std::shared_ptr<RigidBody> rigidbody) {
             raise_if_error(self.AddRigidBody(rigidbody.get(), /*take_ownership=*/false));
           },
           py::keep_alive<1, 2>())
      .def("SetGravity", &WorldModel::SetGravity)
      .def("GetGravity", [](const WorldModel& self) { return Vector3d(self.GetGravity()); })
      .def("EnableGravityEstimation", &WorldModel::EnableGravityEstimation)
      .def("NumberOfLandmarks", &WorldModel::NumberOfLandmarks)
      .def("NumberOfRigidBodies", &WorldModel::NumberOfRigidBodies);

  // ---- solver options / summary (the fields calico.cpp:352-395 binds, plus the HIP backend's own) ----
  py::class_<Summary>(m, "Summary")
      .def("BriefReport", &Summary::BriefReport)
      .def("FullReport", &Summary::FullReport)
      .def("IsSolutionUsable", &Summary::IsSolutionUsable)
      .def_readonly("initial_cost", &Summary::initial_cost)
      .def_readonly("final_cost", &Summary::final_cost)
      .def_readonly("termination_type", &Summary::termination_type)
      .def_readonly("num_iterations", &Summary::num_iterations)
      .def_readonly("num_successful_steps", &Summary::num_successful_steps)
      .def_readonly("num_unsuccessful_steps", &Summary::num_unsuccessful_steps)
      .def_readonly("num_residual_blocks", &Summary::num_residual_blocks)
      .def_readonly("num_residuals", &Summary::num_residuals)
      .def_readonly("num_parameter_blocks", &Summary::num_parameter_blocks)
      .def_readonly("num_parameters", &Summary::num_parameters)
      .def_readonly("num_parameter_blocks_reduced", &Summary::num_parameter_blocks_reduced)
      .def_readonly("num_parameters_reduced", &Summary::num_parameters_reduced)
      .def_readonly("num_effective_parameters_reduced", &Summary::num_effective_parameters_reduced)
      .def_readonly("num_residual_blocks_reduced", &Summary::num_residual_blocks_reduced)
      .def_readonly("num_residuals_reduced", &Summary::num_residuals_reduced)
      .def_readonly("total_time_in_seconds", &Summary::total_time_in_seconds);

  py::class_<SolverOptions>(m, "SolverOptions")
      .def(py::init([] { return DefaultSolverOptions(); }))
      .def_readwrite("max_num_iterations", &SolverOptions::max_num_iterations)
      .def_readwrite("num_threads", &SolverOptions::num_threads)
      .def_readwrite("function_tolerance", &SolverOptions::function_tolerance)
      .def_readwrite("gradient_tolerance", &SolverOptions::gradient_tolerance)
      .def_readwrite("parameter_tolerance", &SolverOptions::parameter_tolerance)
      .def_property(
          "minimizer_progress_to_stdout", [](const SolverOptions& o) { return o.minimizer_progress_to_stdout != 0; },
          [](SolverOptions& o, bool v) { o.minimizer_progress_to_stdout = v ? 1 : 0; })
      .def_readwrite("jacobi_scaling", &SolverOptions::jacobi_scaling)
      .def_readwrite("sync_every", &SolverOptions::sync_every)
      .def_readwrite("initial_trust_region_radius", &SolverOptions::initial_trust_region_radius)
      .def_readwrite("max_trust_region_radius", &SolverOptions::max_trust_region_radius);
  m.def("DefaultSolverOptions", &DefaultSolverOptions);

  py::class_<BatchOptimizer>(m, "BatchOptimizer")
      .def(py::init<>())
      .def("AddSensor", [](BatchOptimizer& self, std::shared_ptr<Sensor> sensor) { self.AddSensor(sensor.get(), /*take_ownership=*/false); },
           py::keep_alive<1, 2>())
      .def("AddTrajectory",
           [](BatchOptimizer& self, std::shared_ptr<Trajectory> trajectory) { self.AddTrajectory(trajectory.get(), /*take_ownership=*/false); },
           py::keep_alive<1, 2>())
      .def("AddWorldModel",
           [](BatchOptimizer& self, std::shared_ptr<WorldModel> world_model) { self.AddWorldModel(world_model.get(), /*take_ownership=*/false); },
           py::keep_alive<1, 2>())
      .def(
          "Optimize",
          [](BatchOptimizer& self, const SolverOptions& options, int device) {
            auto summary = self.Optimize(options, device);
            raise_if_error(summary.status());
            return summary.value();
          },
          py::arg("options") = DefaultSolverOptions(), py::arg("device") = 0);
}
