"""Pins the CPU oracle against the known answers the reference's own tests hold
(SURVEY.md §8c): geometry_test.cpp, bspline_test.cpp, camera_models_test.cpp,
gyroscope/accelerometer_models_test.cpp. CPU only."""
import ctypes as C

import numpy as np
import pytest

import helpers
from calico_amd import synthetic as syn


@pytest.fixture(scope="module")
def lib():
    helpers.build_oracle()
    L = helpers.oracle_lib()
    L.oracle_spline_create.restype = C.c_void_p
    return L


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def mat(fn, *vecs, n=9):
    out = np.zeros(n)
    fn(*[dp(np.ascontiguousarray(v, dtype=float)) for v in vecs], dp(out))
    return out


# ---------------------------------------------------------------- geometry_test.cpp
def test_so3_log_of_exp_small_angles(lib):
    """geometry_test.cpp:27-40: |Ln(Exp(phi)) - phi| < 1e-7 for theta = 1e-12 .. 1e-3."""
    rng = np.random.default_rng(0)
    for i in range(10):
        axis = rng.standard_normal(3)
        axis /= np.linalg.norm(axis)
        phi = (10.0 ** i) * 1e-12 * axis
        R = mat(lib.oracle_exp_so3, phi)
        back = mat(lib.oracle_ln_so3, R, n=3)
        assert np.linalg.norm(back - phi) < 1e-7


def test_exp_so3_is_rotation_and_matches_rodrigues(lib):
    rng = np.random.default_rng(1)
    for _ in range(20):
        phi = rng.uniform(-2, 2, 3)
        R = mat(lib.oracle_exp_so3, phi).reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        q = syn.quat_from_axis_angle(phi)
        v = rng.standard_normal(3)
        assert np.allclose(R @ v, syn.quat_rotate(q, v), atol=1e-13)


def test_rodrigues_jacobian_against_rotation_deltas(lib):
    """geometry_test.cpp:42-161: omega = J(phi) phidot and alpha = Jdot phidot + J phiddot against
    rotation deltas over dt = 1e-6 on the fixture trajectory (1e-4 / 1e-2)."""
    stamps, quats, trans = syn.default_synthetic_poses()
    knots, basis, ctrl = syn.fit_trajectory(stamps, quats, trans)
    dt = 1e-6

    def ln(R):
        return mat(lib.oracle_ln_so3, R.ravel(), n=3)

    def ex(phi):
        return mat(lib.oracle_exp_so3, phi).reshape(3, 3)

    for t in stamps[5:-5:3]:
        p = [syn.spline_eval(knots, basis, ctrl, 6, np.array([t + k * dt]), 0)[0] for k in range(3)]
        for sign in (1.0, -1.0):
            R = [ex(sign * pk[:3]) for pk in p]
            w0 = ln(R[1] @ R[0].T) / dt
            w1 = ln(R[2] @ R[1].T) / dt
            alpha_num = (w1 - w0) / dt
            pd = sign * syn.spline_eval(knots, basis, ctrl, 6, np.array([t]), 1)[0][:3]
            if np.linalg.norm(pd) < 0.05:
                continue  # rotation increments over 1e-6 s below double resolution: the numeric side is noise
            pdd = sign * syn.spline_eval(knots, basis, ctrl, 6, np.array([t]), 2)[0][:3]
            J = mat(lib.oracle_exp_so3_jacobian, sign * p[0][:3]).reshape(3, 3)
            Jd = mat(lib.oracle_exp_so3_jacobian_dot, sign * p[0][:3], pd).reshape(3, 3)
            assert np.linalg.norm(J @ pd - w0) < 1e-4
            assert np.linalg.norm(Jd @ pd + J @ pdd - alpha_num) < 1e-2


def test_rodrigues_hessian_restatement_and_quirk(lib):
    """geometry.h:172-210. Two independent restatements (C++ oracle, numpy) agree. NOTE (quirk Q12):
    the reference's coefficients c0, c2 are NOT those of d(ExpSO3Jacobian)/dphi (c0 belongs to the
    derivative of the rotation matrix, c2 has a sign flipped), so ExpSO3JacobianDot is not the time
    derivative of ExpSO3Jacobian for general motion; it coincides for rotation about a fixed axis,
    which is all the reference's own test exercises. Parity means reproducing the formula as written."""
    rng = np.random.default_rng(2)
    for _ in range(10):
        phi = rng.uniform(-1.5, 1.5, 3)
        H = mat(lib.oracle_exp_so3_hessian, phi, n=27).reshape(3, 3, 3)
        assert np.abs(H - syn.exp_so3_hessian(phi)).max() < 1e-14
        J = mat(lib.oracle_exp_so3_jacobian, phi).reshape(3, 3)
        assert np.abs(J - syn.exp_so3_jacobian(phi)).max() < 1e-14
        # fixed-axis motion: Jdot(phi, s*phi) * phi == 0 contribution structure -> Jdot*phidot finite & equal
        # to the directional derivative of J along phi applied to phidot
        s = 0.37
        h = 1e-6
        dJ = (mat(lib.oracle_exp_so3_jacobian, phi * (1 + h)) - mat(lib.oracle_exp_so3_jacobian, phi * (1 - h))).reshape(3, 3) / (2 * h)
        Jd = mat(lib.oracle_exp_so3_jacobian_dot, phi, s * phi).reshape(3, 3)
        assert np.abs(Jd @ (s * phi) - s * dJ @ (s * phi)).max() < 1e-7


def test_angle_axis_to_quaternion(lib):
    q = mat(lib.oracle_angle_axis_to_quaternion, [0.0, 0.0, 0.0], n=4)
    assert np.array_equal(q, [1, 0, 0, 0])
    q = mat(lib.oracle_angle_axis_to_quaternion, [np.pi / 2, 0.0, 0.0], n=4)
    assert np.allclose(q, [np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0], atol=1e-15)


# ---------------------------------------------------------------- bspline_test.cpp
def _fit_test_spline(lib):
    t = 0.1 * np.arange(101)
    data = np.zeros((101, 6))
    data[:, 0], data[:, 1], data[:, 2] = np.cos(t), np.sin(1.5 * t), t * np.cos(t)
    s = C.c_void_p(lib.oracle_spline_create())
    assert lib.oracle_spline_fit_vectors(s, 101, dp(t), dp(data), C.c_double(5.0), 6) == 0
    return s, t


def test_bspline_interpolation_precision(lib):
    """bspline_test.cpp:52-94: derivatives 0..3 within 1e-6, 1e-5, 1e-4, 1e-2."""
    s, t = _fit_test_spline(lib)
    ti = (t[-1] - t[0]) / 201 * np.arange(201)
    expect = [
        np.stack([np.cos(ti), np.sin(1.5 * ti), ti * np.cos(ti)], 1),
        np.stack([-np.sin(ti), 1.5 * np.cos(1.5 * ti), np.cos(ti) - ti * np.sin(ti)], 1),
        np.stack([-np.cos(ti), -2.25 * np.sin(1.5 * ti), -2.0 * np.sin(ti) - ti * np.cos(ti)], 1),
        np.stack([np.sin(ti), -3.375 * np.cos(1.5 * ti), ti * np.sin(ti) - 3.0 * np.cos(ti)], 1),
    ]
    for d, tol in enumerate([1e-6, 1e-5, 1e-4, 1e-2]):
        out = np.zeros((201, 6))
        assert lib.oracle_spline_interpolate(s, 201, dp(ti), d, dp(out)) == 0
        assert np.abs(out[:, :3] - expect[d]).max() < tol


def test_bspline_invalid_arguments(lib):
    """bspline_test.cpp:34-50: derivative -1 / == order and time -1 -> kInvalidArgument (3)."""
    s, _ = _fit_test_spline(lib)
    out = np.zeros(6)
    zero, neg = np.array([0.0]), np.array([-1.0])
    assert lib.oracle_spline_interpolate(s, 1, dp(zero), -1, dp(out)) == 3
    assert lib.oracle_spline_interpolate(s, 1, dp(zero), 6, dp(out)) == 3
    assert lib.oracle_spline_interpolate(s, 1, dp(neg), 0, dp(out)) == 3


def test_uniform_basis_matrices_golden(lib):
    """exact-rational evaluation of the Qin recursion (bspline.hpp:191-244)."""
    m6 = np.array([[1, 26, 66, 26, 1, 0], [-5, -50, 0, 50, 5, 0], [10, 20, -60, 20, 10, 0], [-10, 20, 0, -20, 10, 0],
                   [5, -20, 30, -20, 5, 0], [-1, 5, -10, 10, -5, 1]]) / 120.0
    m4 = np.array([[1, 4, 1, 0], [-3, 0, 3, 0], [3, -6, 3, 0], [-1, 3, -3, 1]]) / 6.0
    for order, gold in ((6, m6), (4, m4)):
        t = 0.1 * np.arange(60)
        data = np.zeros((60, 6))
        s = C.c_void_p(lib.oracle_spline_create())
        assert lib.oracle_spline_fit_vectors(s, 60, dp(t), dp(data), C.c_double(5.0), order) == 0
        o, nk, nc, ns = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        lib.oracle_spline_sizes(s, C.byref(o), C.byref(nk), C.byref(nc), C.byref(ns))
        assert nc.value == nk.value - order and ns.value == nk.value - 2 * (order - 1) - 1
        B = np.zeros((ns.value, order, order))
        lib.oracle_spline_get(s, None, dp(B), None)
        assert np.abs(B - gold).max() < 5e-14  # floating-point recursion per segment (1e-16-level differences)
        assert np.abs(syn.basis_matrices(syn.knot_vector(0.0, t[-1], order, 5.0), order) - gold).max() < 5e-14


def test_spline_index_semantics(lib):
    """bspline.hpp:138-150: upper_bound - 1; last valid knot -> size-2; beyond -> -1."""
    s, t = _fit_test_spline(lib)
    lib.oracle_spline_index.argtypes = [C.c_void_p, C.c_double]
    assert lib.oracle_spline_index(s, 0.0) == 0
    assert lib.oracle_spline_index(s, 0.2) == 1
    assert lib.oracle_spline_index(s, 0.19999) == 0
    assert lib.oracle_spline_index(s, 10.0) == 49      # == last valid knot: 51 valid knots -> 49
    assert lib.oracle_spline_index(s, 10.0001) == -1
    knots = syn.knot_vector(0.0, 10.0, 6, 5.0)
    assert list(syn.spline_index(knots, 6, np.array([0.0, 0.2, 0.19999, 10.0, 10.0001]))) == [0, 1, 0, 49, -1]


# ---------------------------------------------------------------- camera_models_test.cpp
def _grid_points():
    R = np.diag([1.0, -1.0, -1.0])
    tc = np.array([0.75, 0.75, 1.0])
    n = int(1.5 / 0.025) + 1
    pts = np.array([[i * 0.025, j * 0.025, 0.0] for i in range(n) for j in range(n)])
    return (pts - tc) @ R  # R^T (p - t), R symmetric


def _newton_opencv(k, px, iters=30):
    """The reference's UnprojectPixel for OpenCv5/8 (camera_models.h:157-214, 315-380), restated."""
    f, cx, cy, k1, k2, p1, p2, k3 = k[:8]
    k4, k5, k6 = (k[8:11] if len(k) == 11 else (0.0, 0.0, 0.0))
    xd0, yd0 = (px[:, 0] - cx) / f, (px[:, 1] - cy) / f
    x, y = xd0.copy(), yd0.copy()
    for _ in range(iters):
        x2, y2 = x * x, y * y
        r2 = x2 + y2
        num = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
        den = 1 + r2 * (k4 + r2 * (k5 + r2 * k6))
        s = num / den
        ex = xd0 - (s * x + 2 * p1 * x * y + p2 * (r2 + 2 * x2))
        ey = yd0 - (s * y + 2 * p2 * x * y + p1 * (r2 + 2 * y2))
        dnum = k1 + r2 * (2 * k2 + 3 * r2 * k3)
        dden = k4 + r2 * (2 * k5 + 3 * r2 * k6)
        ds = 2 * (dnum - s * dden) / den
        a = ds * x2 + s + 2 * (p1 * y + 3 * p2 * x)
        b = ds * x * y + 2 * (p1 * x + p2 * y)
        c = ds * y2 + s + 2 * (p2 * x + 3 * p1 * y)
        det = 1.0 / (a * c - b * b)
        x, y = x + det * (c * ex - b * ey), y + det * (-b * ex + a * ey)
    v = np.stack([x, y, np.ones_like(x)], 1)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _unproject_closed_form(model, k, px):
    """The reference's closed-form UnprojectPixel (camera_models.h:674-701, 798-832, 920-945, 1036-1062)."""
    f, cx, cy = k[:3]
    mx, my = (px[:, 0] - cx) / f, (px[:, 1] - cy) / f
    if model == 4:
        xi, al = k[3], k[4]
        r2 = mx * mx + my * my
        mz = (1 - al * al * r2) / (al * np.sqrt(1 - (2 * al - 1) * r2) + 1 - al)
        inv_s = (mz * xi + np.sqrt(mz * mz + (1 - xi * xi) * r2)) / (mz * mz + r2)
        v = np.stack([inv_s * mx, inv_s * my, inv_s * mz - xi], 1)
    elif model == 5:
        w = k[3]
        r = np.sqrt(mx * mx + my * my)
        tt = 2 * np.tan(w / 2)
        eta = np.where(r * r < 1e-5, w / tt, np.sin(r * w) / (np.where(r > 0, r, 1) * tt))
        v = np.stack([eta * mx, eta * my, np.cos(r * w)], 1)
    elif model == 6:
        al = k[3]
        mx, my = (1 - al) * mx, (1 - al) * my
        r2 = mx * mx + my * my
        xi = al / (1 - al)
        s = (xi + np.sqrt(1 + (1 - xi * xi) * r2)) / (1 + r2)
        v = np.stack([s * mx, s * my, s - xi], 1)
    else:
        al, be = k[3], k[4]
        r2 = mx * mx + my * my
        mz = (1 - be * al * al * r2) / (al * np.sqrt(1 - (2 * al - 1) * be * r2) + (1 - al))
        v = np.stack([mx, my, mz], 1)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


CAMERA_CASES = [
    (1, [785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4, -1.853e-2], 1e-10),
    (2, [785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4, -1.853e-2, 1.225e-1, -5.26e-2, 8.58e-3], 1e-10),
    (4, [785, 640, 400, 0.5, 0.5], 1e-12),
    (5, [785, 640, 400, 0.05], 1e-12),
    (6, [785, 640, 400, 0.5], 1e-12),
    (7, [785, 640, 400, 0.5, 0.5], 2e-2),
]


@pytest.mark.parametrize("model,intr,tol", CAMERA_CASES)
def test_camera_model_round_trip(lib, model, intr, tol):
    """camera_models_test.cpp:104-253: unproject(project(p)) == p/|p| on the 61x61 grid, the
    reference's intrinsics and tolerances; the inverse is the reference's own, restated above."""
    k = np.array(intr, float)
    pts = _grid_points()
    px = np.zeros((len(pts), 2))
    for i, p in enumerate(pts):
        assert lib.oracle_project_point(model, dp(k), dp(np.ascontiguousarray(p)), dp(px[i])) == 0
    bearing = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    back = _newton_opencv(k, px) if model in (1, 2) else _unproject_closed_form(model, k, px)
    assert np.abs(back - bearing).max() < tol
    # the vectorised numpy generator used for synthetic data agrees with the oracle
    px_np, valid = syn.project_point(model, k, pts)
    assert valid.all() and np.abs(px_np - px).max() < 1e-9


def test_kannala_brandt_round_trip(lib):
    """camera_models_test.cpp:150-172 (tolerance 1e-9): theta_d inverted by bisection here."""
    k = np.array([785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4], float)
    pts = _grid_points()
    px = np.zeros((len(pts), 2))
    for i, p in enumerate(pts):
        assert lib.oracle_project_point(3, dp(k), dp(np.ascontiguousarray(p)), dp(px[i])) == 0
    m = (px - k[1:3]) / k[0]
    rd = np.linalg.norm(m, axis=1)
    lo, hi = np.zeros_like(rd), np.full_like(rd, np.pi / 2)
    for _ in range(200):
        th = 0.5 * (lo + hi)
        t2 = th * th
        val = th * (1 + t2 * (k[3] + t2 * (k[4] + t2 * (k[5] + t2 * k[6]))))
        lo, hi = np.where(val < rd, th, lo), np.where(val < rd, hi, th)
    th = 0.5 * (lo + hi)
    scale = np.where(rd > 0, np.sin(th) / np.where(rd > 0, rd, 1), 0)
    back = np.stack([m[:, 0] * scale, m[:, 1] * scale, np.cos(th)], 1)
    bearing = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    assert np.abs(back - bearing).max() < 1e-9


def test_projection_validity_rules(lib):
    """camera_models.h:108-111 (z <= 0 invalid), :638, :884, :998."""
    out = np.zeros(2)
    behind = np.array([0.1, 0.1, -1.0])
    for model, intr, _ in CAMERA_CASES[:2] + [(3, [785, 640, 400, -0.3, 0.1, 0, 0], 0), (5, [785, 640, 400, 0.05], 0)]:
        assert lib.oracle_project_point(model, dp(np.array(intr, float)), dp(behind), dp(out)) == 3
    # unified models accept points slightly behind the camera plane (z > -w d)
    assert lib.oracle_project_point(6, dp(np.array([785, 640, 400, 0.5])), dp(np.array([1.0, 0.0, -0.1])), dp(out)) == 0
    assert lib.oracle_project_point(6, dp(np.array([785, 640, 400, 0.5])), dp(np.array([0.0, 0.0, -1.0])), dp(out)) == 3


# ---------------------------------------------------------------- gyroscope/accelerometer_models_test.cpp
@pytest.mark.parametrize("model,intr", [(1, [1.3]), (2, [1.3, 0.01, -0.01, 0.01]),
                                        (3, [1.01, 0.99, 1.02, 1e-3, -2e-3, 1.5e-3, -1e-3, 2e-3, 1e-3, 0.01, -0.01, 0.01])])
def test_imu_model_round_trip(lib, model, intr):
    """Unproject(Project(x)) == x within 1e-9; the inverse restated from gyroscope_models.h:94-100,150-163,237-270."""
    k = np.array(intr, float)
    rng = np.random.default_rng(3)
    for _ in range(20):
        w = rng.standard_normal(3)
        f = np.zeros(3)
        assert lib.oracle_imu_project(model, dp(k), dp(w), dp(f)) == 0
        if model == 1:
            back = f / k[0]
        elif model == 2:
            back = (f - k[1:4]) / k[0]
        else:
            A = np.array([[1, k[3], k[4]], [k[5], 1, k[6]], [k[7], k[8], 1]])
            back = np.linalg.solve(np.diag(k[:3]) @ A, f - k[9:12])
        assert np.abs(back - w).max() < 1e-9
        assert np.abs(syn.imu_project(model, k, w[None])[0] - f).max() < 1e-14
    assert lib.oracle_camera_num_params(1) == 8 and lib.oracle_camera_num_params(7) == 5
    assert lib.oracle_imu_num_params(3) == 12 and lib.oracle_imu_num_params(0) == -1


def test_blocked_cholesky_equals_plain_cholesky():
    """Beyond 1500 unknowns (long trajectories) the oracle factors its dense normal equations with a blocked, threaded
    Cholesky; it must give what the plain loop gives (only the order of the additions differs)."""
    import ctypes as C
    lib = helpers.oracle_lib()
    fn = lib.oracle_cholesky_blocked_vs_plain
    fn.restype = C.c_double
    fn.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32]
    rng = np.random.default_rng(7)
    for n, threads in ((1, 1), (63, 1), (64, 4), (65, 3), (200, 8), (517, 5)):
        M = rng.standard_normal((n, n + 5))
        A = np.ascontiguousarray(M @ M.T + 1e-3 * np.eye(n))      # SPD, condition number ~1e3-1e5
        b = rng.standard_normal(n)
        d = fn(n, A.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), threads)
        assert 0.0 <= d <= 1e-10, (n, threads, d)
    A = -np.eye(3)
    assert fn(3, A.ctypes.data_as(C.POINTER(C.c_double)), np.ones(3).ctypes.data_as(C.POINTER(C.c_double)), 2) == -1.0
