"""The reference's special-case branches, driven on the device against the oracle (`-m gpu`), on crafted inputs that the
noisy generic scenes of test_gpu_parity.py never hit:
  * KannalaBrandt with a point 5e-10 off the optical axis (the r < 1e-9 branch, camera_models.h:444-446),
  * FieldOfView with w^2 < 1e-5 and with r^2 < 1e-5 (camera_models.h:762-772),
  * rotation vectors of norm 0 and 1e-8 in the control points (the Taylor branches of ExpSO3 / its Jacobian, geometry.h),
  * a stamp exactly on the last valid knot (GetSplineIndex, bspline.hpp:138-150),
  * a latency that moves the evaluation time across a knot while the segment stays frozen at the raw stamp
    (quirk Q3, camera_cost_functor.cpp:13-14,52 vs camera_cost_functor.h:114-119) -- Jacobian included,
  * a candidate point at which a projection cannot be evaluated MID-solve (the step is rejected with cost
    1.797693e+308, like the four rows of the reference's own Ceres table),
and BASELINE.json configs[0] (the 500-block plumbing scene) as a whole.
Everything is compared through evaluate() -- [cost, J^T r, J^T J] to 1e-9 -- residuals, and full solves."""
import numpy as np
import pytest

import helpers
from calico_amd import _capi, synthetic as syn

pytestmark = pytest.mark.gpu


def _camera(model, intr, meas, stamps, pidx, latency=0.0, free_latency=False, free_extrinsics=False):
    return syn.SensorSpec(_capi.SENSOR_CAMERA, model, "cam", np.asarray(intr, float), np.array([0.0, 0, 0, 1.0]), np.zeros(3),
                          latency, np.asarray(intr, float), np.array([0.0, 0, 0, 1.0]), np.zeros(3), latency, True,
                          free_extrinsics, free_latency, sigma=1.0, loss=0, loss_scale=1.0, meas=np.asarray(meas, float),
                          stamps=np.asarray(stamps, float), point_idx=np.asarray(pidx, np.int32))


def _scene(ctrl_fn, points, sensors_fn, t1=1.0, knot_frequency=10.0, order=6):
    knots = syn.knot_vector(0.0, t1, order, knot_frequency)
    basis = syn.basis_matrices(knots, order)
    n_ctrl = len(knots) - order
    ctrl = np.array([ctrl_fn(i) for i in range(n_ctrl)], float)
    sc = syn.Scene(order, knots, basis, ctrl.copy(), ctrl.copy(), np.asarray(points, float), np.array([0.0, 0, 0, 1.0]), np.zeros(3),
                   np.array([0.0, 0.0, -9.80665]), [])
    sc.sensors = sensors_fn(sc)
    return sc


def _compare_evaluation(hip, oracle, scene, rtol=1e-9):
    g, r = syn.build_problem(hip, scene), syn.build_problem(oracle, scene)
    cg, gg, Hg = g.problem.evaluate()
    cr, gr, Hr = r.problem.evaluate()
    assert np.isfinite(cr) and np.all(np.isfinite(gr)) and np.all(np.isfinite(Hr))
    assert abs(cg - cr) <= rtol * max(abs(cr), 1e-300)
    assert np.abs(gg - gr).max() <= rtol * max(np.abs(gr).max(), 1e-300)
    # entries are compared relative to the geometric mean of their diagonal entries; a control point whose B-spline
    # weight is zero up to rounding (t on a knot) has a diagonal of ~1e-29, which is a zero, not a scale
    sg = np.sqrt(np.diag(Hr))
    assert np.all(np.abs(Hg - Hr) <= rtol * np.outer(sg, sg) + 1e-13 * np.abs(Hr).max())
    for i, s in enumerate(scene.sensors):
        rg, vg = g.problem.residuals(g.sensor_ids[i], s.n, s.dim)
        rr, vr = r.problem.residuals(r.sensor_ids[i], s.n, s.dim)
        assert np.array_equal(vg, vr)
        assert np.abs(rg - rr).max() <= rtol * max(1.0, np.abs(rr).max())
    return g, r


def _static_points():
    # camera at the origin looking down +z (all poses identity): the first point sits 5e-10 off the optical axis, inside
    # the r < 1e-9 branch. (EXACTLY on the axis the reference's autodiff differentiates sqrt(x^2 + y^2) at 0 and hands
    # Ceres a NaN Jacobian -- the oracle does the same; that input is a failed evaluation, not a branch to mirror.)
    return np.array([[5e-10, 0.0, 1.0], [0.1, 0.0, 1.0], [0.0, -0.2, 1.5], [0.3, 0.25, 2.0], [-0.2, 0.1, 0.8], [1e-4, -2e-4, 1.0],
                     [-0.4, -0.3, 1.2], [0.05, 0.4, 1.1]])


def _obs_all_points(scene, stamps, model, intr, noise=0.3, seed=5, **kw):
    rng = np.random.default_rng(seed)
    spl = (scene.knots, scene.basis, scene.ctrl_true, scene.order)
    px, valid, st, _, pidx = syn.project_camera(spl, model, np.asarray(intr, float), np.array([0.0, 0, 0, 1.0]), np.zeros(3), 0.0,
                                                 np.asarray(stamps, float), scene.points, scene.body_q, scene.body_t)
    assert valid.all()
    return [_camera(model, intr, px + noise * rng.standard_normal(px.shape), st, pidx, **kw)]


@pytest.mark.parametrize("theta", [0.0, 1e-8])
def test_kannala_brandt_on_axis_and_zero_rotation(hip, oracle, theta):
    """r < 1e-9 (the near-axis point; with theta = 1e-8 it moves out to r ~ 1e-8, the regular branch right next to the
    switch) together with control points whose rotation vector is exactly 0 / of norm 1e-8."""
    kb = syn._TRUE_INTRINSICS[3]
    scene = _scene(lambda i: [theta, 0.0, 0.0, 0.0, 0.0, 0.0], _static_points(),
                   lambda sc: _obs_all_points(sc, [0.05, 0.31, 0.52, 0.77], 3, kb))
    # theta = 1e-8 puts the point at r ~ 1e-8, just outside the branch: there the reference's autodiff differentiates
    # theta_d / r by the quotient rule and loses eight digits to cancellation (the oracle does the same), the device's
    # closed form does not -- the comparison is held to what the reference itself resolves
    _compare_evaluation(hip, oracle, scene, rtol=1e-9 if theta == 0.0 else 1e-7)


@pytest.mark.parametrize("w", [1e-3, 0.9])
def test_field_of_view_small_w_and_small_r(hip, oracle, w):
    """w^2 < 1e-5 switches the distortion off; with a regular w the points with r^2 < 1e-5 take the limit branch."""
    intr = np.array([600.0, 640.0, 400.0, w])
    scene = _scene(lambda i: [0.0, 0.01 * i, 0.0, 0.0, 0.0, 0.001 * i], _static_points(),
                   lambda sc: _obs_all_points(sc, [0.05, 0.31, 0.52, 0.77], 5, intr))
    _compare_evaluation(hip, oracle, scene)


def test_stamp_on_the_last_valid_knot(hip, oracle):
    """t == last valid knot belongs to the last segment (bspline.hpp:138-150): evaluated at u = 1 of that segment."""
    cv = syn._TRUE_INTRINSICS[1]

    def sensors(sc):
        last_valid = sc.knots[len(sc.knots) - (sc.order - 1) - 1]
        return _obs_all_points(sc, [0.3, last_valid - 0.05, last_valid], 1, cv)
    scene = _scene(lambda i: [0.02 * np.sin(i), 0.01 * i, -0.015 * i, 0.01 * i, 0.0, 0.02 * np.cos(i)], _static_points(), sensors)
    last_valid = scene.knots[len(scene.knots) - (scene.order - 1) - 1]
    assert scene.sensors[0].stamps.max() == last_valid
    _compare_evaluation(hip, oracle, scene)


def test_latency_moves_the_evaluation_time_across_a_knot(hip, oracle):
    """Stamps 1 ms after a knot, latency 4 ms (free): the pose is evaluated BEFORE the knot with the control points and
    basis of the segment the raw stamp falls into (Q3). Residuals and the whole Jacobian (latency column included)."""
    cv = syn._TRUE_INTRINSICS[1]

    def sensors(sc):
        knot_times = sc.knots[sc.order + 1: sc.order + 5]
        s = _obs_all_points(sc, knot_times + 1e-3, 1, cv, latency=0.0, free_latency=True, free_extrinsics=True)[0]
        s.latency = 4e-3
        return [s]
    scene = _scene(lambda i: [0.03 * np.sin(0.7 * i), 0.02 * np.cos(i), -0.01 * i, 0.02 * i, 0.01 * np.sin(i), 0.03 * np.cos(0.5 * i)],
                   _static_points(), sensors)
    seg_raw = [syn.spline_index(scene.knots, scene.order, t) for t in scene.sensors[0].stamps]
    seg_eval = [syn.spline_index(scene.knots, scene.order, t - 4e-3) for t in scene.sensors[0].stamps]
    assert all(a == b + 1 for a, b in zip(seg_raw, seg_eval))       # the evaluation time sits in the segment before
    g, r = _compare_evaluation(hip, oracle, scene)
    # and a few LM iterations from there walk the same path
    o = hip.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 5
    sg, sr = g.problem.solve(o), r.problem.solve(o)
    assert [i.step_is_successful for i in g.problem.iterations()] == [i.step_is_successful for i in r.problem.iterations()]
    np.testing.assert_allclose([i.cost for i in g.problem.iterations()], [i.cost for i in r.problem.iterations()], rtol=1e-7)
    assert sg.num_iterations == sr.num_iterations


def test_candidate_that_cannot_be_evaluated_mid_solve(hip, oracle):
    """A point one centimetre in front of the camera, the start three centimetres back: after three ordinary iterations
    a step pushes the point behind the camera, the candidate's cost cannot be evaluated, the step is rejected with
    cost 1.797693e+308 and the radius goes down the /2 /4 ladder -- rows 1-4 of the reference's Ceres table, here in the
    middle of a solve (found with the oracle; iterations 4 and 5)."""
    cv = syn._TRUE_INTRINSICS[1].copy()
    pts = _static_points().copy()
    pts[5] = [0.002, -0.001, 0.01]
    scene = _scene(lambda i: [0.0, 0.0, 0.0, 0.0, 0.0, 0.0], pts,
                   lambda sc: _obs_all_points(sc, [0.05, 0.31, 0.52, 0.77], 1, cv, noise=0.0))
    scene.ctrl = scene.ctrl + np.array([0.0, 0.0, 0.0, 0.0, 0.0, -0.03])
    built = {}
    logs = {}
    for name, api in (("hip", hip), ("oracle", oracle)):
        b = syn.build_problem(api, scene)
        o = api.default_options()
        o.minimizer_progress_to_stdout = 0
        o.max_num_iterations = 15
        b.problem.solve(o)
        built[name] = b
        logs[name] = [(i.iteration, i.step_is_valid, i.step_is_successful, i.cost, i.trust_region_radius) for i in b.problem.iterations()]
    dbl_max = np.finfo(np.float64).max
    invalid_rows = [row[0] for row in logs["oracle"] if row[3] == dbl_max]
    assert invalid_rows and min(invalid_rows) >= 2, logs["oracle"]     # the scene does what it was built for: MID-solve
    assert not any(row[2] for row in logs["oracle"][1:max(invalid_rows) + 1])    # a ladder of rejections, evaluable or not,
    assert any(row[2] for row in logs["oracle"][max(invalid_rows) + 1:])         # and accepted steps once the radius is small
    assert len(logs["hip"]) == len(logs["oracle"])
    for a, b in zip(logs["hip"], logs["oracle"]):
        assert a[:3] == b[:3], (logs["hip"], logs["oracle"])
        assert a[3] == b[3] if b[3] == dbl_max else abs(a[3] - b[3]) <= 1e-6 * abs(b[3])
        assert abs(a[4] - b[4]) <= 1e-9 * b[4]


def test_config0_plumbing_scene(hip, oracle):
    """BASELINE.json configs[0]: one pinhole camera, ~500 reprojection residual blocks."""
    scene = syn.config_scene(0)
    assert scene.num_blocks == 500
    g, r = _compare_evaluation(hip, oracle, scene)
    o = hip.default_options()
    o.minimizer_progress_to_stdout = 0
    sg, sr = g.problem.solve(o), r.problem.solve(o)
    assert sg.termination_type == sr.termination_type == _capi.CONVERGENCE
    assert sg.num_iterations == sr.num_iterations
    eg, er = syn.read_back(g, scene)[0], syn.read_back(r, scene)[0]
    for a, b in zip(eg, er):
        np.testing.assert_allclose(a["intrinsics"], b["intrinsics"], rtol=1e-6, atol=1e-9)
    for i, s in enumerate(scene.sensors):
        assert np.array_equal(g.problem.inlier_mask(g.sensor_ids[i], s.n, 3.0), r.problem.inlier_mask(r.sensor_ids[i], s.n, 3.0))
