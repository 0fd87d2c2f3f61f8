"""Edge cases of the path on the GPU (`-m gpu`), each against the oracle's behaviour on the same input:
sensors without observations, tiny and ragged problems, a residual block that cannot be evaluated at the
starting point (camera.cpp:75 -> Internal in the reference; Ceres fails the solve), and one that becomes invalid
at a candidate point (the step is rejected)."""
import copy

import numpy as np
import pytest

import helpers
from calico_amd import _capi, synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(**kw):
    args = dict(cam_rate=10.0, imu_rate=50.0, duration=3.0, segment_duration=3.0 / 23.9, pixel_noise=0.1,
                gyro_noise=1e-3, accel_noise=1e-2, seed=3)
    args.update(kw)
    return syn.make_scene(2, 1, True, 2, **args)


def _solve(api, scene, max_iter=30):
    built = syn.build_problem(api, scene)
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = max_iter
    return built, built.problem.solve(o)


def _compare(hip, oracle, scene, max_iter=30, est_rtol=1e-6):
    g, sg = _solve(hip, scene, max_iter)
    r, sr = _solve(oracle, scene, max_iter)
    assert sg.termination_type == sr.termination_type, (sg.message, sr.message)
    assert sg.num_iterations == sr.num_iterations
    assert sg.num_residual_blocks == sr.num_residual_blocks
    if sr.termination_type != _capi.FAILURE:
        assert abs(sg.final_cost - sr.final_cost) <= 1e-8 * max(sr.final_cost, 1e-300)
        eg, er = syn.read_back(g, scene)[0], syn.read_back(r, scene)[0]
        for a, b in zip(eg, er):
            for key in ("intrinsics", "t", "q"):
                np.testing.assert_allclose(a[key], b[key], rtol=est_rtol, atol=1e-9)
    return g, r, sg, sr


def test_sensor_without_observations(hip, oracle):
    """A registered gyroscope with zero measurements contributes parameters but no residual blocks."""
    scene = _scene()
    scene = copy.deepcopy(scene)
    for s in scene.sensors:
        if s.kind == _capi.SENSOR_GYROSCOPE:
            s.meas = s.meas[:0]
            s.stamps = s.stamps[:0]
    _compare(hip, oracle, scene)


def test_tiny_problem(hip, oracle):
    """A handful of observations: cells with a single block, frames of a few points."""
    scene = _scene(max_cam_obs=24, imu_rate=20.0, duration=1.0, segment_duration=0.5)
    _compare(hip, oracle, scene, max_iter=10)


def test_ragged_frames(hip, oracle):
    """Frames of very different sizes (every third observation of one camera dropped, a tail cut off)."""
    scene = copy.deepcopy(_scene())
    cam = [s for s in scene.sensors if s.kind == _capi.SENSOR_CAMERA][1]
    keep = np.ones(cam.n, bool)
    keep[::3] = False
    keep[-37:] = False
    cam.meas, cam.stamps, cam.point_idx = cam.meas[keep], cam.stamps[keep], cam.point_idx[keep]
    if cam.is_outlier is not None:
        cam.is_outlier = cam.is_outlier[keep]
    _compare(hip, oracle, scene)


def test_block_invalid_at_start_fails_the_solve(hip, oracle):
    """A chart point behind the camera at the initial estimate: the residual cannot be evaluated, Ceres gives up at
    iteration 0 (FAILURE); both backends must say so and leave the parameters alone."""
    scene = copy.deepcopy(_scene())
    scene.points[5] = scene.points[5] + np.array([0.0, 0.0, 50.0])     # far behind every camera
    g, r, sg, sr = _compare(hip, oracle, scene)
    assert sr.termination_type == _capi.FAILURE
    eg = syn.read_back(g, scene)[0]
    for a, s in zip(eg, scene.sensors):
        np.testing.assert_array_equal(a["intrinsics"], s.intrinsics)


def test_outlier_tagging_loop_on_device(hip, oracle):
    """The demos' loop (kalibr_multicam_demo.ipynb:636-677): solve, tag ||r|| > tau, solve again -- on the device via
    calico_mark_outliers, against the oracle doing it the reference's way (new problem without the tagged ids,
    camera.cpp:121-124, started from the first solution). Tags bit-exact, estimates within 1e-6."""
    tau = 3.0
    scene = _scene(outlier_fraction=0.04, seed=21)
    g = syn.build_problem(hip, scene)
    r = syn.build_problem(oracle, scene)
    o = hip.default_options(); o.minimizer_progress_to_stdout = 0; o.max_num_iterations = 40
    oo = oracle.default_options(); oo.minimizer_progress_to_stdout = 0; oo.max_num_iterations = 40
    sg1, sr1 = g.problem.solve(o), r.problem.solve(oo)
    assert sg1.termination_type == sr1.termination_type == _capi.CONVERGENCE
    cams = [i for i, s in enumerate(scene.sensors) if s.kind == _capi.SENSOR_CAMERA]
    # pass 1: tags on the device vs the oracle's inlier mask
    scene2 = copy.deepcopy(scene)
    est, ctrl = syn.read_back(r, scene)
    n_tagged = 0
    for i in cams:
        s = scene.sensors[i]
        marked = g.problem.mark_outliers(g.sensor_ids[i], tau)
        ref_inlier = r.problem.inlier_mask(r.sensor_ids[i], s.n, tau).astype(bool)
        assert marked == int((~ref_inlier).sum())
        assert np.array_equal(g.problem.inlier_mask(g.sensor_ids[i], s.n, tau).astype(bool), ref_inlier)   # bit-exact
        assert (~ref_inlier)[s.is_outlier].mean() > 0.9       # the gross outliers are among the tagged
        n_tagged += marked
        s2 = scene2.sensors[i]
        s2.meas, s2.stamps, s2.point_idx = s.meas[ref_inlier], s.stamps[ref_inlier], s.point_idx[ref_inlier]
        s2.is_outlier = s.is_outlier[ref_inlier]
    assert n_tagged > 0
    # second solve: device continues on the tagged problem; the oracle gets a new problem from its first solution
    for s2, e in zip(scene2.sensors, est):
        s2.intrinsics, s2.t, s2.q, s2.latency = e["intrinsics"].copy(), e["t"].copy(), e["q"].copy(), float(e["latency"])
    scene2.ctrl = ctrl.copy()
    r2 = syn.build_problem(oracle, scene2)
    sg2, sr2 = g.problem.solve(o), r2.problem.solve(oo)
    assert sg2.num_residual_blocks == sr2.num_residual_blocks == sg1.num_residual_blocks - n_tagged
    assert sg2.termination_type == sr2.termination_type
    assert abs(sg2.final_cost - sr2.final_cost) <= 1e-6 * sr2.final_cost
    eg, _ = syn.read_back(g, scene)
    er, _ = syn.read_back(r2, scene2)
    for a, b in zip(eg, er):
        for key in ("intrinsics", "t", "q"):
            np.testing.assert_allclose(a[key], b[key], rtol=1e-6, atol=1e-9)
    # clearing the tags restores the full problem
    for i in cams:
        g.problem.set_outlier_mask(g.sensor_ids[i], None)
    assert g.problem.solve(o).num_residual_blocks == sg1.num_residual_blocks


def test_residual_heatmap_on_device(hip):
    """utils.py:12-50 on the device: binned RMSE / feature counts equal the host-side restatement on the residuals the
    same problem hands back; tagged observations are left out."""
    from calico_amd import calico
    scene = syn.make_scene(2, 1, False, cam_rate=10.0, duration=3.0, segment_duration=3.0 / 23.9, pixel_noise=0.5, seed=9)
    built = syn.build_problem(hip, scene)
    P = built.problem
    sid, sensor = built.sensor_ids[1], scene.sensors[1]
    mask = np.zeros(sensor.n, np.uint8)
    mask[::7] = 1                                   # tag every seventh observation
    P.set_outlier_mask(sid, mask)
    rmse, count = P.residual_heatmap(sid, 1280, 800, 8, 12)
    res, valid = P.residuals(sid, sensor.n, 2, check=False)
    pairs = []
    for i in range(sensor.n):
        if mask[i] or not valid[i]:
            continue
        m = calico.CameraMeasurement()
        m.pixel = sensor.meas[i]
        pairs.append((m, res[i]))
    _, ref_rmse, ref_count = calico.ComputeRmseHeatmapAndFeatureCount(pairs, 1280, 800, 8, 12)
    assert np.array_equal(count, ref_count.astype(np.int64))
    filled = ref_count > 0
    assert filled.sum() > 20
    np.testing.assert_allclose(rmse[filled], ref_rmse[filled], rtol=1e-12)
    assert np.isnan(rmse[~filled]).all()
    again, _ = P.residual_heatmap(sid, 1280, 800, 8, 12)
    assert np.array_equal(again[filled], rmse[filled])          # fixed-order reduction
