"""The oracle's cost functors, loss handling and LM loop against what the reference's tests pin
(gyroscope_test.cpp:159-183, accelerometer_test.cpp:179-203, batch_optimizer_test.cpp:32-213) and
against self-verification (dual-number Jacobians vs finite differences). CPU only."""
import ctypes as C

import numpy as np
import pytest

import helpers
from calico_amd import _capi, synthetic as syn


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def lib():
    helpers.build_oracle()
    L = helpers.oracle_lib()
    L.oracle_spline_create.restype = C.c_void_p
    L.oracle_num_residuals.restype = C.c_int64
    return L


def _oracle_spline(lib, knots, basis, ctrl):
    """An oracle BSpline holding exactly the given control points."""
    stamps, quats, trans = syn.default_synthetic_poses()
    s = C.c_void_p(lib.oracle_spline_create())
    qw = np.ascontiguousarray(quats[:, [3, 0, 1, 2]])
    assert lib.oracle_spline_fit_poses(s, len(stamps), dp(stamps), dp(qw), dp(np.ascontiguousarray(trans)),
                                       C.c_double(10.0), 6) == 0
    assert lib.oracle_spline_set_ctrl(s, dp(np.ascontiguousarray(ctrl))) == 0
    return s


def _adopt_oracle_spline_tables(lib, spl, scene):
    """Use the oracle's own knot vector / basis matrices in the problem (the numpy and C++ recursions
    differ in the last bit), as the reference uses ONE spline object for Project and the functors."""
    o, nk, nc, ns = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    lib.oracle_spline_sizes(spl, C.byref(o), C.byref(nk), C.byref(nc), C.byref(ns))
    K = np.zeros(nk.value)
    B = np.zeros((ns.value, o.value, o.value))
    lib.oracle_spline_get(spl, dp(K), dp(B), None)
    assert np.abs(K - scene.knots).max() == 0.0 and np.abs(B - scene.basis).max() < 1e-15
    scene.knots, scene.basis = K, B


@pytest.mark.parametrize("model", [1, 2, 3])
def test_perfect_data_perfect_residuals_imu(lib, oracle, model):
    """gyroscope_test.cpp:159-183 / accelerometer_test.cpp:179-203: Project -> AddMeasurements ->
    AddResidualsToProblem -> Evaluate gives cost == 0.0 EXACTLY and 3 residuals per block."""
    scene = syn.make_scene(0, 1, True, model, perturb=False)
    spl = _oracle_spline(lib, scene.knots, scene.basis, scene.ctrl)
    _adopt_oracle_spline_tables(lib, spl, scene)
    times = syn.default_synthetic_poses()[0]
    for s in scene.sensors:
        meas = np.zeros((len(times), 3))
        st = np.zeros(len(times))
        if s.kind == _capi.SENSOR_GYROSCOPE:
            rc = lib.oracle_project_gyroscope(spl, model, dp(s.intrinsics), dp(s.q), C.c_double(s.latency), len(times),
                                              dp(times), dp(meas), dp(st))
        else:
            rc = lib.oracle_project_accelerometer(spl, model, dp(s.intrinsics), dp(s.q), dp(s.t), C.c_double(s.latency),
                                                  dp(scene.gravity), len(times), dp(times), dp(meas), dp(st))
        assert rc == 0
        # the numpy generator used for benchmarks agrees with the reference-style generator
        assert np.abs(meas - s.meas).max() < 1e-10 and np.array_equal(st, s.stamps)
        # the reference test leaves the latency at its default 0, so that stamp - latency == stamp exactly
        s.meas, s.stamps, s.latency = meas, times.copy(), 0.0
    built = syn.build_problem(oracle, scene)
    # cost only, as the reference test's problem.Evaluate(..., &cost, nullptr, nullptr, nullptr): the
    # functor runs in plain double (Jet division multiplies by a reciprocal, which is not bit-identical)
    cost = C.c_double(-1.0)
    assert lib.oracle_evaluate_cost(built.problem.h, 1, C.byref(cost)) == 0
    assert cost.value == 0.0
    assert lib.oracle_num_residuals(built.problem.h) == 3 * scene.num_blocks


@pytest.mark.parametrize("model", [1, 2, 3, 4, 5, 6, 7])
def test_perfect_data_camera(lib, oracle, model):
    """Camera::Project (camera.cpp:155-208) -> functor: residuals vanish (the two paths compose the
    rotations in a different order, so 'zero' is to rounding, not exact)."""
    scene = syn.make_scene(2, model, False, perturb=False)
    spl = _oracle_spline(lib, scene.knots, scene.basis, scene.ctrl)
    times = syn.default_synthetic_poses()[0]
    for s in scene.sensors:
        T, P = len(times), len(scene.points)
        px = np.zeros((T * P, 2))
        valid = np.zeros(T * P, np.uint8)
        st = np.zeros(T * P)
        assert lib.oracle_project_camera(spl, model, dp(s.intrinsics), dp(s.q), dp(s.t), C.c_double(s.latency), T, dp(times),
                                         P, dp(np.ascontiguousarray(scene.points)), dp(scene.body_q), dp(scene.body_t),
                                         dp(px), valid.ctypes.data_as(C.POINTER(C.c_uint8)), dp(st)) == 0
        v = valid.astype(bool)
        assert v.sum() == s.n  # visibility rule z > 0 (camera_test.cpp)
        assert np.abs(px[v] - s.meas).max() < 1e-8 and np.array_equal(st[v], s.stamps)
    built = syn.build_problem(oracle, scene)
    cost, _, _ = built.problem.evaluate(want_jtj=False)
    assert cost < 1e-16


def _fd_gradient_check(lib, built, order_blocks):
    """With a loss function the corrected Jacobian is not d(corrected residual)/dx (Triggs), but
    J'^T r' must still be the gradient of the robustified cost 1/2 sum rho."""
    P = built.problem
    _, g, _ = P.evaluate(want_jtj=False)

    def cost():
        c = C.c_double(0)
        assert lib.oracle_evaluate_cost(P.h, 1, C.byref(c)) == 0
        return c.value

    def qplus(x, d):
        nd = np.linalg.norm(d)
        return syn.quat_mul(np.concatenate([np.sin(nd) / nd * d, [np.cos(nd)]]), x)

    col, worst = 0, 0.0
    gs = np.abs(g).max()
    for bid, sz, man in order_blocks:
        x0 = P.get_param_block(bid, sz)
        tsz = 3 if man else sz
        for c in range(tsz):
            h = 1e-6 * max(1.0, 1.0 if man else abs(x0[c]))
            d = np.zeros(tsz)
            d[c] = h
            P.set_param_block(bid, qplus(x0, d) if man else x0 + d)
            cp = cost()
            P.set_param_block(bid, qplus(x0, -d) if man else x0 - d)
            cm = cost()
            P.set_param_block(bid, x0)
            worst = max(worst, abs((cp - cm) / (2 * h) - g[col]) / max(1.0, abs(g[col]), 1e-6 * gs))
            col += 1
    assert col == len(g)
    return worst


def _fd_check(lib, built, scene, order_blocks):
    P = built.problem
    n = P.num_effective_parameters()
    nres = lib.oracle_num_residuals(P.h)
    J = np.zeros((nres, n))
    r0 = np.zeros(nres)
    assert lib.oracle_evaluate_jacobian(P.h, dp(r0), dp(J)) == 0

    def resid():
        r = np.zeros(nres)
        assert lib.oracle_evaluate_jacobian(P.h, dp(r), None) == 0
        return r

    def qplus(x, d):
        nd = np.linalg.norm(d)
        return syn.quat_mul(np.concatenate([np.sin(nd) / nd * d, [np.cos(nd)]]), x)

    col, worst = 0, 0.0
    for bid, sz, man in order_blocks:
        x0 = P.get_param_block(bid, sz)
        tsz = 3 if man else sz
        for c in range(tsz):
            h = 1e-6 * max(1.0, 1.0 if man else abs(x0[c]))
            d = np.zeros(tsz)
            d[c] = h
            P.set_param_block(bid, qplus(x0, d) if man else x0 + d)
            rp = resid()
            P.set_param_block(bid, qplus(x0, -d) if man else x0 - d)
            rm = resid()
            P.set_param_block(bid, x0)
            fd = (rp - rm) / (2 * h)
            worst = max(worst, np.abs(fd - J[:, col]).max() / max(1.0, np.abs(J[:, col]).max()))
            col += 1
    assert col == n
    return worst


@pytest.mark.parametrize("camera_model,imu_model,robust", [(1, 2, False), (2, 3, True), (3, 1, False), (4, 2, False),
                                                           (5, 2, True), (6, 3, False), (7, 2, False)])
def test_dual_number_jacobian_vs_finite_differences(lib, oracle, camera_model, imu_model, robust):
    """Self-verification of the 'parity unpinned' part: Jet-style derivatives, manifold projection and
    loss correction against central differences of the residuals."""
    scene = syn.make_scene(2, camera_model, True, imu_model, robust=robust, pixel_noise=0.5, gyro_noise=0.05,
                           accel_noise=0.5, free_chart_pose=True, seed=5)
    rng = np.random.default_rng(0)
    for s in scene.sensors:
        keep = np.sort(rng.choice(s.n, 25, replace=False))
        s.meas, s.stamps = s.meas[keep], s.stamps[keep]
        if s.point_idx is not None:
            s.point_idx = s.point_idx[keep]
    built = syn.build_problem(oracle, scene)
    segs = np.concatenate([syn.spline_index(scene.knots, scene.order, s.stamps) for s in scene.sensors])
    used = np.zeros(len(scene.ctrl), bool)
    for sg in segs:
        used[sg:sg + 6] = True
    blocks = [(int(built.ctrl_blocks[i]), 6, 0) for i in range(len(scene.ctrl)) if used[i]]
    others = [(built.body_t_block, 3, 0), (built.body_q_block, 4, 1)]
    for s, sb in zip(scene.sensors, built.sensor_blocks):
        if s.enable_intrinsics:
            others.append((sb["intrinsics"], len(s.intrinsics), 0))
        if s.enable_extrinsics:
            others += [(sb["t"], 3, 0), (sb["q"], 4, 1)]
        if s.enable_latency:
            others.append((sb["latency"], 1, 0))
    if robust:
        worst = _fd_gradient_check(lib, built, blocks + sorted(others))
        assert worst < 1e-4
    else:
        worst = _fd_check(lib, built, scene, blocks + sorted(others))
        assert worst < 2e-5


def test_loss_functions_against_closed_forms(oracle):
    """ceres HuberLoss / CauchyLoss (optimization_utils.h:31-47): cost = 1/2 sum rho(|r|^2)."""
    scene = syn.make_scene(1, 1, True, 2, pixel_noise=0.3, gyro_noise=0.05, accel_noise=0.5, outlier_fraction=0.1, seed=9)
    plain = syn.build_problem(oracle, scene)
    res = [plain.problem.residuals(plain.sensor_ids[i], s.n, s.dim)[0] for i, s in enumerate(scene.sensors)]
    for loss, scale in ((1, 0.7), (2, 1.3)):
        for s in scene.sensors:
            s.loss, s.loss_scale = loss, scale
        built = syn.build_problem(oracle, scene)
        cost, _, _ = built.problem.evaluate(want_jtj=False)
        expect = 0.0
        for r in res:
            sq = (r * r).sum(1)
            b = scale * scale
            rho = np.where(sq > b, 2 * scale * np.sqrt(sq) - b, sq) if loss == 1 else b * np.log1p(sq / b)
            expect += 0.5 * rho.sum()
        assert abs(cost - expect) <= 1e-12 * expect
        # robustified residual write-back is NOT affected (apply_loss_function=false, camera.cpp:73)
        again = built.problem.residuals(built.sensor_ids[0], scene.sensors[0].n, 2)[0]
        assert np.array_equal(again, res[0])


def test_toy_stereo_camera_and_imu_calibration(oracle):
    """batch_optimizer_test.cpp:32-213 restated: 2 OpenCv5 cameras + ScaleAndBias gyro/accel, perfect
    data, perturbed start; CONVERGENCE, final cost < 1e-7, every estimate within 1e-7 of truth,
    Ceres' default 50 iterations."""
    scene = syn.make_scene(2, 1, True, 2, seed=4)
    built = syn.build_problem(oracle, scene)
    o = oracle.default_options()
    o.minimizer_progress_to_stdout = 0
    o.num_threads = 8
    s = built.problem.solve(o)
    assert s.termination_type == _capi.CONVERGENCE
    assert s.final_cost < 1e-7
    est, _ = syn.read_back(built, scene)
    for e, sp in zip(est, scene.sensors):
        assert np.abs(e["intrinsics"] - sp.intrinsics_true).max() < 1e-7
        assert np.abs(e["t"] - sp.t_true).max() < 1e-7
        assert np.abs(e["q"] - sp.q_true).max() < 1e-7
        assert abs(e["latency"] - sp.latency_true) < 1e-7
    assert s.num_residual_blocks == 2 * 8640 + 2 * 240 and s.num_residuals == 2 * 2 * 8640 + 2 * 3 * 240
    # world model (36 points, pose, gravity) + 185 control points + 4 sensors x (intr, t, q, latency)
    assert s.num_parameter_blocks == 36 + 3 + 185 + 16
    assert s.num_effective_parameters_reduced == 185 * 6 + 8 + 8 + 7 + 4 + 4 + 4 + 7 + 3  # gyro lever arm has no effect but is a free block


def test_error_conventions(oracle):
    """Status codes of the boundary: absl numbering."""
    P = _capi.Problem(oracle)
    with pytest.raises(_capi.CalicoError) as e:
        P.add_param_block([1, 2, 3], _capi.MANIFOLD_EIGEN_QUATERNION)
    assert e.value.code == _capi.INVALID_ARGUMENT
    b = P.add_param_block(np.zeros(8))
    q = P.add_param_block([0, 0, 0, 1], _capi.MANIFOLD_EIGEN_QUATERNION)
    t = P.add_param_block(np.zeros(3))
    lat = P.add_param_block([0.0])
    with pytest.raises(_capi.CalicoError) as e:   # camera.cpp:94-97: model not set
        P.add_sensor(_capi.SENSOR_CAMERA, 0, b, q, t, lat)
    assert e.value.code == _capi.FAILED_PRECONDITION
    with pytest.raises(_capi.CalicoError) as e:   # wrong intrinsics size
        P.add_sensor(_capi.SENSOR_CAMERA, 3, b, q, t, lat)
    assert e.value.code == _capi.INVALID_ARGUMENT
    sid = P.add_sensor(_capi.SENSOR_CAMERA, 1, b, q, t, lat)
    with pytest.raises(_capi.CalicoError) as e:   # residuals before the trajectory
        P.add_camera_residuals(sid, np.zeros((1, 2)), np.zeros(1), np.zeros(1, np.int32), np.zeros(1, np.int32))
    assert e.value.code == _capi.FAILED_PRECONDITION
