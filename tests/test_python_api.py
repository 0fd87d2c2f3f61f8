"""`from calico_amd import calico`: the reference's Python surface (calico/calico.cpp bindings + calico/utils.py)
over the HIP backend. The CPU part restates calico/test/python_bindings_test.py (accessors, status codes, model
parameter counts) and calico/test/python_utils_test.py (detections -> measurements, Zhang initialiser); the GPU
part runs the notebooks' call sequence end to end: synthesize with Project(), perturb, BatchOptimizer.Optimize(),
read residual pairs, tag outliers, re-solve."""
import copy

import numpy as np
import pytest

import __graft_entry__ as _entry

_entry.build_hip()
_entry.build_python_module()   # no-ops when the in-tree libraries are up to date

from calico_amd import calico, synthetic as syn  # noqa: E402

ROT = [-0.774982, -0.1549964, -0.2324946, 0.5668556]
POS = [1.5, 2.3, 6.8]


def test_pose3d_accessors():  # python_bindings_test.py:9-16
    pose = calico.Pose3d()
    pose.rotation = ROT
    pose.translation = POS
    np.testing.assert_allclose(pose.rotation, np.array(ROT), 1e-7)
    np.testing.assert_equal(pose.translation, np.array(POS))
    clone = copy.deepcopy(pose)
    clone.translation = [0, 0, 0]
    np.testing.assert_equal(pose.translation, np.array(POS))


@pytest.mark.parametrize("cls,model,meas_cls", [
    (calico.Accelerometer, calico.AccelerometerIntrinsicsModel.kAccelerometerScaleOnly, calico.AccelerometerMeasurement),
    (calico.Gyroscope, calico.GyroscopeIntrinsicsModel.kGyroscopeScaleOnly, calico.GyroscopeMeasurement),
])
def test_imu_sensor_accessors(cls, model, meas_cls):  # python_bindings_test.py:18-98
    sensor = cls()
    sensor.SetName("test")
    assert sensor.GetName() == "test"
    assert sensor.SetModel(model).ok()
    assert sensor.GetModel() == model
    sensor.SetIntrinsics([1])
    np.testing.assert_equal([1], sensor.GetIntrinsics())
    ext = calico.Pose3d()
    ext.rotation = ROT
    ext.translation = POS
    sensor.SetExtrinsics(ext)
    np.testing.assert_allclose(ext.rotation, sensor.GetExtrinsics().rotation, 1e-7)
    np.testing.assert_equal(ext.translation, sensor.GetExtrinsics().translation)
    assert sensor.SetLatency(0.02).ok()
    assert sensor.GetLatency() == 0.02
    m = meas_cls()
    m.id.stamp = 0
    m.id.sequence = 0
    assert sensor.AddMeasurement(m).ok()
    dup = sensor.AddMeasurement(m)
    assert not dup.ok() and dup.code() == calico.StatusCode.kInvalidArgument
    ms = []
    for i in range(3):
        mi = meas_cls()
        mi.id.stamp = i + 1
        mi.id.sequence = i + 1
        ms.append(mi)
    assert sensor.AddMeasurements(ms).ok()
    assert sensor.NumberOfMeasurements() == 4
    assert not sensor.SetMeasurementNoise(0.0).ok() and sensor.SetMeasurementNoise(0.1).ok()


def test_camera_accessors():  # python_bindings_test.py:100-139, camera_test.cpp:82-101
    camera = calico.Camera()
    with pytest.raises(RuntimeError, match="Error: "):
        camera.SetIntrinsics(np.arange(8.0))          # model not set yet (camera.cpp:24-27)
    assert camera.SetModel(calico.CameraIntrinsicsModel.kOpenCv5).ok()
    with pytest.raises(RuntimeError):
        camera.SetIntrinsics([1, 2, 3])                # wrong size
    camera.SetIntrinsics([1, 2, 3, 4, 5, 6, 7, 8])
    np.testing.assert_equal(np.arange(1.0, 9.0), camera.GetIntrinsics())
    m = calico.CameraMeasurement()
    m.pixel = [3.0, 4.0]
    m.id.stamp = 0.5
    m.id.image_id = 2
    m.id.feature_id = 7
    assert camera.AddMeasurement(m).ok()
    assert camera.AddMeasurement(m).code() == calico.StatusCode.kInvalidArgument
    other = calico.CameraMeasurement(m)
    other.id.feature_id = 8
    st = camera.AddMeasurements([m, other])            # Q11: duplicates are reported, unique ones still added
    assert st.code() == calico.StatusCode.kInvalidArgument
    table = camera.GetMeasurementIdToMeasurement()
    assert len(table) == 2 and m.id in table
    np.testing.assert_equal(table[m.id].pixel, [3.0, 4.0])
    assert "feature_id: 7" in str(m.id)
    with pytest.raises(RuntimeError):
        unknown = calico.CameraObservationId()
        unknown.feature_id = 99
        camera.MarkOutlierById(unknown)
    camera.MarkOutlierById(m.id)


def test_model_parameter_counts():  # python_bindings_test.py:222-257
    cams = [("kOpenCv5", 8), ("kOpenCv8", 11), ("kKannalaBrandt", 7), ("kDoubleSphere", 5), ("kFieldOfView", 4),
            ("kUnifiedCamera", 4), ("kExtendedUnifiedCamera", 5)]
    for name, n in cams:
        c = calico.Camera()
        c.SetModel(getattr(calico.CameraIntrinsicsModel, name))
        c.SetIntrinsics(np.random.rand(n))
        with pytest.raises(RuntimeError):
            c.SetIntrinsics(np.random.rand(n + 1))
    for cls, enum, prefix in ((calico.Accelerometer, calico.AccelerometerIntrinsicsModel, "kAccelerometer"),
                              (calico.Gyroscope, calico.GyroscopeIntrinsicsModel, "kGyroscope")):
        for name, n in (("ScaleOnly", 1), ("ScaleAndBias", 4), ("VectorNav", 12)):
            s = cls()
            s.SetModel(getattr(enum, prefix + name))
            s.SetIntrinsics(np.random.rand(n))


def test_rigid_body_and_world_model():  # python_bindings_test.py:146-184, world_model_test.cpp
    definition = {0: [0, 0, 0], 1: [1, 1, 1], 2: [2, 2, 2]}
    body = calico.RigidBody()
    body.model_definition = dict(definition)
    body.id = 1
    body.world_pose_is_constant = True
    body.model_definition_is_constant = True
    for k, p in body.model_definition.items():
        np.testing.assert_equal(definition[k], p)
    assert body.id == 1 and body.world_pose_is_constant and body.model_definition_is_constant
    world = calico.WorldModel()
    world.AddRigidBody(body)
    with pytest.raises(RuntimeError):
        world.AddRigidBody(body)
    assert world.NumberOfRigidBodies() == 1
    np.testing.assert_allclose(world.GetGravity(), [0, 0, -9.80665])
    lm = calico.Landmark()
    lm.point = [1, 2, 3]
    lm.id = 4
    world.AddLandmark(lm)
    assert world.NumberOfLandmarks() == 1


def test_detections_to_camera_measurements():  # python_utils_test.py:11-24
    detections = {i: np.array([float(i), float(i)]) for i in range(600)}
    measurements = calico.DetectionsToCameraMeasurements(detections, 1.0, 32)
    assert len(measurements) == len(detections)
    for m in measurements:
        np.testing.assert_equal(detections[m.id.feature_id], m.pixel)
        assert m.id.stamp == 1.0 and m.id.image_id == 32 and m.id.model_id == 0


def test_initialize_pinhole_and_poses():  # python_utils_test.py:26-89
    from scipy.spatial.transform import Rotation as R
    true_intrinsics = [400, 410, 10, 100, 250]
    K = np.array([[400, 10, 100], [0, 410, 250], [0, 0, 1.0]])
    R_cw = [R.from_rotvec(v).as_matrix() for v in ([np.pi, np.pi / 3, 0], [np.pi, -np.pi / 3, 0], [np.pi, np.pi / 12, 0],
                                                   [np.pi + np.pi / 12, 0, 0], [np.pi, np.pi / 6, np.pi / 12])]
    t_cw = [np.array(t) for t in ([0.5, 0.5, 1], [0.6, 0.6, 1.25], [0.5, 0.5, 0.75], [0.4, 0.4, 1.1], [0.5, 0.6, 0.9])]
    world = np.array([[0.1 * x, 0.1 * y, 0.0, 1.0] for x in range(11) for y in range(11)]).T
    model = {i: world[:3, i].copy() for i in range(world.shape[1])}
    detections = []
    for Rc, tc in zip(R_cw, t_cw):
        pr = K @ (np.hstack((Rc, tc.reshape(3, 1))) @ world)
        detections.append({j: np.array([pr[0, j] / pr[2, j], pr[1, j] / pr[2, j]]) for j in range(world.shape[1])})
    intrinsics, R_wc, t_wc = calico.InitializePinholeAndPoses(detections, model)
    np.testing.assert_almost_equal(true_intrinsics, intrinsics, decimal=3)
    for aR, at, eR, et in zip(R_wc, t_wc, R_cw, t_cw):
        np.testing.assert_almost_equal(-R.from_matrix(aR).as_rotvec(), R.from_matrix(eR).as_rotvec(), decimal=6)
        np.testing.assert_almost_equal(-aR.T @ at, et, decimal=5)


def test_rmse_heatmap_and_feature_count():  # utils.py:12-50
    pairs = []
    for (u, v, r) in ((10.0, 10.0, [3.0, 4.0]), (12.0, 8.0, [0.0, 0.0]), (630.0, 470.0, [1.0, 0.0])):
        m = calico.CameraMeasurement()
        m.pixel = [u, v]
        pairs.append((m, np.array(r)))
    image, rmse, count = calico.ComputeRmseHeatmapAndFeatureCount(pairs, 640, 480, num_rows=8, num_cols=12)
    assert image.shape == (480, 640) and rmse.shape == (8, 12)
    assert count[0, 0] == 2 and count[7, 11] == 1 and count.sum() == 3
    assert rmse[0, 0] == pytest.approx(np.sqrt(25.0 / 2)) and rmse[7, 11] == 1.0
    assert image[0, 0] == rmse[0, 0] and image[479, 639] == rmse[7, 11] and np.isnan(rmse[3, 3])


# ---------------------------------------------------------------------------------------------------------------
def _poses():
    stamps, quats, trans = syn.default_synthetic_poses()
    out = {}
    for t, q, p in zip(stamps, quats, trans):
        pose = calico.Pose3d()
        pose.rotation = [q[3], q[0], q[1], q[2]]
        pose.translation = p
        out[float(t)] = pose
    return stamps, out


@pytest.mark.gpu
def test_trajectory_fit_and_interpolate():  # python_bindings_test.py:141-144, trajectory_test.cpp:23-34
    trajectory = calico.Trajectory()
    trajectory.FitSpline({0.0: calico.Pose3d(), 1.0: calico.Pose3d()})
    stamps, poses = _poses()
    trajectory.FitSpline(poses)
    got = trajectory.Interpolate(list(stamps))
    for t, g in zip(stamps, got):
        e = poses[float(t)]
        assert min(np.abs(g.rotation - e.rotation).max(), np.abs(g.rotation + e.rotation).max()) < 1e-3
        assert np.abs(g.translation - e.translation).max() < 1e-3
    with pytest.raises(RuntimeError, match="Error: "):
        trajectory.Interpolate([-1.0])


@pytest.mark.gpu
def test_batch_optimizer_stub():  # python_bindings_test.py:186-219: sensors without measurements
    accelerometer = calico.Accelerometer()
    assert accelerometer.SetModel(calico.AccelerometerIntrinsicsModel.kAccelerometerScaleOnly).ok()
    accelerometer.SetIntrinsics([1])
    gyroscope = calico.Gyroscope()
    assert gyroscope.SetModel(calico.GyroscopeIntrinsicsModel.kGyroscopeScaleOnly).ok()
    gyroscope.SetIntrinsics([1])
    camera = calico.Camera()
    assert camera.SetModel(calico.CameraIntrinsicsModel.kOpenCv5).ok()
    camera.SetIntrinsics([1, 2, 3, 4, 5, 6, 7, 8])
    trajectory = calico.Trajectory()
    trajectory.FitSpline({0.0: calico.Pose3d(), 1.0: calico.Pose3d()})
    optimizer = calico.BatchOptimizer()
    optimizer.AddSensor(accelerometer)
    optimizer.AddSensor(gyroscope)
    optimizer.AddSensor(camera)
    optimizer.AddTrajectory(trajectory)
    optimizer.AddWorldModel(calico.WorldModel())
    options = calico.DefaultSolverOptions()
    options.minimizer_progress_to_stdout = False
    summary = optimizer.Optimize(options)
    assert summary.num_residual_blocks == 0


@pytest.mark.gpu
def test_camera_calibration_with_outlier_tagging():
    """The notebook flow (kalibr_multicam_demo.ipynb cells around 636-677): optimise, look at the residuals, tag the
    gross outliers, optimise again."""
    stamps, poses = _poses()
    trajectory = calico.Trajectory()
    trajectory.FitSpline(poses)
    chart = calico.RigidBody()
    chart.model_definition = {i: p for i, p in enumerate(syn.planar_points())}
    chart.world_pose_is_constant = True
    chart.model_definition_is_constant = True
    world = calico.WorldModel()
    world.AddRigidBody(chart)
    truth = np.array([785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4, -1.853e-2])
    true_camera = calico.Camera()
    true_camera.SetModel(calico.CameraIntrinsicsModel.kOpenCv5)
    true_camera.SetIntrinsics(truth)
    measurements = true_camera.Project([float(t) for t in stamps[::4]], trajectory, world)
    assert len(measurements) > 1000
    rng = np.random.default_rng(5)
    corrupted = set(rng.choice(len(measurements), 12, replace=False).tolist())
    for i in corrupted:
        m = measurements[i]
        m.pixel = m.pixel + np.array([40.0, -35.0])
    camera = calico.Camera()
    camera.SetModel(calico.CameraIntrinsicsModel.kOpenCv5)
    init = 1.01 * truth
    init[3:] = 0.0
    camera.SetIntrinsics(init)
    camera.EnableIntrinsicsEstimation(True)
    camera.EnableExtrinsicsEstimation(False)
    camera.EnableLatencyEstimation(False)
    assert camera.AddMeasurements(measurements).ok()
    optimizer = calico.BatchOptimizer()
    optimizer.AddSensor(camera)
    optimizer.AddTrajectory(trajectory)
    optimizer.AddWorldModel(world)
    options = calico.DefaultSolverOptions()
    options.minimizer_progress_to_stdout = False
    summary = optimizer.Optimize(options)
    assert summary.IsSolutionUsable() and summary.num_residual_blocks == len(measurements)
    assert "Iterations" in summary.BriefReport()
    pairs = camera.GetMeasurementResidualPairs()
    assert len(pairs) == len(measurements)
    norms = np.array([np.linalg.norm(r) for _, r in pairs])
    tagged = [m.id for (m, r), n in zip(pairs, norms) if n > 10.0]
    assert len(tagged) == len(corrupted)
    camera.MarkOutliersById(tagged)
    summary = optimizer.Optimize(options)
    assert summary.num_residual_blocks == len(measurements) - len(corrupted)
    assert summary.final_cost < 1e-6
    np.testing.assert_allclose(camera.GetIntrinsics(), truth, rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_toy_stereo_camera_and_imu_calibration_python():
    """batch_optimizer_test.cpp:32-213 (ToyStereoCameraAndImuCalibration) through the Python surface: perfect
    measurements from Project(), perturbed start, every sensor parameter recovered to 1e-7."""
    stamps, poses = _poses()
    times = [float(t) for t in stamps]
    trajectory = calico.Trajectory()
    trajectory.FitSpline(poses)
    chart = calico.RigidBody()
    chart.model_definition = {i: p for i, p in enumerate(syn.planar_points())}
    chart.world_pose_is_constant = True
    chart.model_definition_is_constant = True
    world = calico.WorldModel()
    world.AddRigidBody(chart)

    def pose(axis, angle_deg, t):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        half = 0.5 * np.deg2rad(angle_deg)
        p = calico.Pose3d()
        p.rotation = [np.cos(half), *(np.sin(half) * axis)]
        p.translation = t
        return p

    true_cam = np.array([785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4, -1.853e-2])
    true_imu = np.array([1.3, 0.01, -0.01, 0.01])
    ex_right = pose([0.68, -0.21, 0.57], 2.0, 0.05 * np.array([0.6, -0.33, 0.54]))
    ex_gyro = pose([-0.44, 0.11, -0.05], 2.0, [0, 0, 0])
    ex_acc = pose([0.26, -0.27, 0.9], 2.0, [0, 0, 0])
    truth = {}
    specs = [("left", calico.Camera, calico.CameraIntrinsicsModel.kOpenCv5, true_cam, calico.Pose3d(), 0.0),
             ("right", calico.Camera, calico.CameraIntrinsicsModel.kOpenCv5, true_cam, ex_right, 0.01),
             ("gyro", calico.Gyroscope, calico.GyroscopeIntrinsicsModel.kGyroscopeScaleAndBias, true_imu, ex_gyro, 0.02),
             ("acc", calico.Accelerometer, calico.AccelerometerIntrinsicsModel.kAccelerometerScaleAndBias, true_imu, ex_acc, 0.02)]
    optimizer = calico.BatchOptimizer()
    sensors = {}
    for name, cls, model, intrinsics, extrinsics, latency in specs:
        true_sensor = cls()
        assert true_sensor.SetModel(model).ok()
        true_sensor.SetIntrinsics(intrinsics)
        true_sensor.SetExtrinsics(extrinsics)
        assert true_sensor.SetLatency(latency).ok()
        measurements = true_sensor.Project(times, trajectory, world)
        assert len(measurements) > 0
        sensor = cls()
        sensor.SetName(name)
        assert sensor.SetModel(model).ok()
        init = 1.01 * intrinsics
        if cls is calico.Camera:
            init[3:] = 0.0
        sensor.SetIntrinsics(init)
        init_ext = calico.Pose3d(extrinsics)
        if name == "right":
            init_ext.translation = extrinsics.translation + 0.01 * np.array([0.3, -0.8, 0.5])
        if name == "acc":
            init_ext.translation = extrinsics.translation + 0.05 * np.array([-0.2, 0.7, 0.4])
        sensor.SetExtrinsics(init_ext)
        sensor.EnableIntrinsicsEstimation(True)
        sensor.EnableExtrinsicsEstimation(name != "left")
        sensor.EnableLatencyEstimation(name != "left")
        assert sensor.AddMeasurements(measurements).ok()
        optimizer.AddSensor(sensor)
        sensors[name] = sensor
        truth[name] = (intrinsics, extrinsics, latency)
    optimizer.AddTrajectory(trajectory)
    optimizer.AddWorldModel(world)
    options = calico.DefaultSolverOptions()
    options.minimizer_progress_to_stdout = False
    options.max_num_iterations = 100
    summary = optimizer.Optimize(options)
    assert summary.IsSolutionUsable() and summary.final_cost < 1e-7
    for name, sensor in sensors.items():
        intrinsics, extrinsics, latency = truth[name]
        np.testing.assert_allclose(sensor.GetIntrinsics(), intrinsics, rtol=0, atol=1e-7)
        got = sensor.GetExtrinsics()
        assert min(np.abs(got.rotation - extrinsics.rotation).max(), np.abs(got.rotation + extrinsics.rotation).max()) < 1e-7
        if name != "gyro":     # the gyroscope's lever arm is unobservable (zero Jacobian): it keeps its start value
            np.testing.assert_allclose(got.translation, extrinsics.translation, rtol=0, atol=1e-7)
        assert abs(sensor.GetLatency() - latency) < 1e-7
