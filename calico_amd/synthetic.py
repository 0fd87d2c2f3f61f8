"""Synthetic calibration problems (numpy, vectorised).

Host-side data generation for tests and benchmarks: the trajectory pattern of
the reference's DefaultSyntheticTest fixture (calico/test_utils.h:11-116), a
uniform B-spline in Qin's general-matrix form (calico/bspline.hpp), and the
synthetic measurement generators Sensor::Project (camera.cpp:155-208,
gyroscope.cpp:56-82, accelerometer.cpp:76-123). Nothing here is on the
optimisation path; it only builds inputs (and is cross-checked against the
CPU oracle in tests/).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _capi

CAMERA_NUM_PARAMS = {1: 8, 2: 11, 3: 7, 4: 5, 5: 4, 6: 4, 7: 5}
IMU_NUM_PARAMS = {1: 1, 2: 4, 3: 12}


# ----------------------------------------------------------------------------
# rotations (quaternions stored x,y,z,w like Eigen coeffs())
# ----------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = np.moveaxis(np.asarray(a, float), -1, 0)
    bx, by, bz, bw = np.moveaxis(np.asarray(b, float), -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_conj(q):
    q = np.asarray(q, float)
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_rotate(q, v):
    q = np.asarray(q, float)
    v = np.asarray(v, float)
    u = q[..., :3]
    uv = 2.0 * np.cross(u, v)
    return v + q[..., 3:4] * uv + np.cross(u, uv)


def quat_from_axis_angle(phi):
    """ceres::AngleAxisToQuaternion; returns x,y,z,w."""
    phi = np.asarray(phi, float)
    th2 = np.sum(phi * phi, -1)
    th = np.sqrt(th2)
    safe = np.where(th2 > 0, th, 1.0)
    k = np.where(th2 > 0, np.sin(0.5 * safe) / safe, 0.5)
    w = np.where(th2 > 0, np.cos(0.5 * safe), 1.0)
    return np.concatenate([phi * k[..., None], w[..., None]], -1)


def quat_to_axis_angle(q):
    """Eigen::AngleAxisd(Quaterniond) -> axis*angle."""
    q = np.asarray(q, float)
    n = np.linalg.norm(q[..., :3], axis=-1)
    ang = 2.0 * np.arctan2(n, np.abs(q[..., 3]))
    s = np.where(q[..., 3] < 0, -1.0, 1.0) / np.where(n > 0, n, 1.0)
    return q[..., :3] * (s * ang)[..., None] * (n > 0)[..., None]


def skew(v):
    v = np.asarray(v, float)
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def exp_so3_jacobian(phi):
    """calico/geometry.h:137-161 (vectorised)."""
    phi = np.asarray(phi, float)
    th = np.linalg.norm(phi, axis=-1)
    safe = np.where(th > 0, th, 1.0)
    hat = phi / safe[..., None]
    K = skew(hat)
    a = (1.0 - np.cos(safe)) / safe
    b = (safe - np.sin(safe)) / safe
    J = np.eye(3) + a[..., None, None] * K + b[..., None, None] * (K @ K)
    return np.where((th > 0)[..., None, None], J, np.eye(3))


def exp_so3_hessian(phi):
    """calico/geometry.h:172-210; returns H[..., i, :, :]."""
    phi = np.asarray(phi, float)
    th = np.linalg.norm(phi, axis=-1)
    safe = np.where(th > 0, th, 1.0)
    it = 1.0 / safe
    hat = phi * it[..., None]
    K = skew(hat)
    ct, st = np.cos(safe), np.sin(safe)
    c0 = ct - st * it
    c1 = (1.0 - ct) * it * it
    c2 = 3.0 * it * it * st - it * (ct - 2.0)
    c3 = it * it * (safe - st)
    G = skew(np.eye(3))
    H = []
    for i in range(3):
        Hi = ((c0 * hat[..., i])[..., None, None] * K + c1[..., None, None] * G[i]
              + (c2 * hat[..., i])[..., None, None] * (K @ K) + c3[..., None, None] * (G[i] @ K + K @ G[i]))
        H.append(np.where((th > 0)[..., None, None], Hi, 0.0))
    return np.stack(H, -3)


# ----------------------------------------------------------------------------
# B-spline (calico/bspline.hpp)
# ----------------------------------------------------------------------------
def knot_vector(t0, t1, order, knot_frequency):
    """bspline.hpp:163-180."""
    deg = order - 1
    nvalid = 1 + int(np.ceil((t1 - t0) * knot_frequency))
    nk = nvalid + 2 * deg
    dt = 1.0 / knot_frequency
    return np.array([t0 + dt * i for i in range(-deg, nk - deg)])


def _basis_matrix(knots, k, i):
    """bspline.hpp:191-244 (Qin recursion) for segment knot index i."""
    if k == 1:
        return np.array([[1.0]])
    Mk1 = _basis_matrix(knots, k - 1, i)
    M1 = np.vstack([Mk1, np.zeros((1, k - 1))])
    M2 = np.vstack([np.zeros((1, k - 1)), Mk1])
    A = np.zeros((k - 1, k))
    B = np.zeros((k - 1, k))
    for index in range(k - 1):
        j = i - k + 2 + index
        den = knots[j + k - 1] - knots[j]
        d0 = 0.0 if den <= 0 else (knots[i] - knots[j]) / den
        d1 = 0.0 if den <= 0 else (knots[i + 1] - knots[i]) / den
        A[index, index], A[index, index + 1] = 1.0 - d0, d0
        B[index, index], B[index, index + 1] = -d1, d1
    return M1 @ A + M2 @ B


def basis_matrices(knots, order):
    deg = order - 1
    nseg = len(knots) - 2 * deg - 1
    return np.stack([_basis_matrix(knots, order, i + deg) for i in range(nseg)])


def spline_index(knots, order, t):
    """bspline.hpp:138-150 (vectorised); -1 beyond the last valid knot."""
    deg = order - 1
    vk = knots[deg:len(knots) - deg]
    t = np.asarray(t, float)
    idx = np.searchsorted(vk, t, side="right") - 1
    idx = np.where(t == vk[-1], len(vk) - 2, idx)
    return np.where(t > vk[-1], -1, idx).astype(np.int32)


def spline_weights(knots, basis, order, t, derivative, seg=None):
    """U·M for each time: (n, order). bspline.hpp:39-72."""
    t = np.asarray(t, float)
    if seg is None:
        seg = spline_index(knots, order, t)
    deg = order - 1
    k0 = knots[seg + deg]
    k1 = knots[seg + deg + 1]
    dt_inv = 1.0 / (k1 - k0)
    u = (t - k0) * dt_inv
    U = np.zeros(t.shape + (order,))
    for i in range(derivative, order):
        coeff = 1.0
        for j in range(i - derivative, i):
            coeff *= (j + 1)
        U[..., i] = coeff * u ** (i - derivative) * dt_inv ** derivative
    return np.einsum("ni,nij->nj", U, basis[seg]), seg


def spline_eval(knots, basis, ctrl, order, t, derivative=0, seg=None):
    W, seg = spline_weights(knots, basis, order, t, derivative, seg)
    idx = seg[:, None] + np.arange(order)[None, :]
    return np.einsum("nj,njc->nc", W, ctrl[idx])


def spline_fit(stamps, data, order, knot_frequency):
    """bspline.hpp:19-37,246-297 (least squares; initialisation only)."""
    stamps = np.asarray(stamps, float)
    knots = knot_vector(stamps[0], stamps[-1], order, knot_frequency)
    basis = basis_matrices(knots, order)
    ncp = len(knots) - order
    W, seg = spline_weights(knots, basis, order, stamps, 0)
    X = np.zeros((len(stamps), ncp))
    rows = np.arange(len(stamps))[:, None]
    X[rows, seg[:, None] + np.arange(order)[None, :]] = W
    # The fixture's design matrix is rank deficient at the trajectory end (fewer
    # samples than control points): minimum-norm least squares keeps the spline bounded.
    ctrl = np.linalg.lstsq(X, data, rcond=1e-9)[0]
    return knots, basis, ctrl


def unwrap_phase_log_map(phi):
    """trajectory.cpp:81-93."""
    phi = np.array(phi, float)
    for i in range(1, len(phi)):
        th = np.linalg.norm(phi[i])
        if th == 0:
            continue
        k = np.round((phi[i] @ phi[i - 1] - th * th) / (2.0 * np.pi * th))
        phi[i] *= (1.0 + 2.0 * np.pi * k / th)
    return phi


def default_synthetic_poses(segment_duration=0.75, samples_per_segment=10, repeats=1):
    """DefaultSyntheticTest (test_utils.h:13-83): stamps, quats (x,y,z,w), positions.

    `repeats` tiles the 24-segment excitation pattern in time."""
    deg = np.pi / 180.0
    q0 = quat_mul(quat_from_axis_angle([0, 0, np.pi]), quat_from_axis_angle([np.pi, 0, 0]))
    t0 = np.array([0.0, 0.0, 1.0])
    ang = [0.0, 30 * deg, 0.0, -30 * deg, 0.0]
    pos = [0.0, 0.5, 0.0, -0.5, 0.0]
    dti = 1.0 / samples_per_segment
    dta = dti * segment_duration
    interp = [(np.sin(dti * i * np.pi - np.pi / 2) + 1.0) / 2.0 for i in range(samples_per_segment)]
    stamps, quats, trans = [], [], []
    t = 0.0
    for _ in range(repeats):
        for axis in np.eye(3):
            for i in range(1, len(ang)):
                for s in interp:
                    th = (ang[i] - ang[i - 1]) * s + ang[i - 1]
                    quats.append(quat_mul(q0, quat_from_axis_angle(axis * th)))
                    trans.append(t0.copy())
                    stamps.append(t)
                    t += dta
            for i in range(1, len(pos)):
                for s in interp:
                    p = (pos[i] - pos[i - 1]) * s + pos[i - 1]
                    quats.append(q0.copy())
                    trans.append(axis * p + t0)
                    stamps.append(t)
                    t += dta
    return np.array(stamps), np.array(quats), np.array(trans)


def fit_trajectory(stamps, quats_xyzw, trans, knot_frequency=10.0, order=6):
    """Trajectory::FitSpline (trajectory.cpp:14-49)."""
    o = np.argsort(stamps)
    stamps, quats_xyzw, trans = stamps[o], quats_xyzw[o], trans[o]
    phi = unwrap_phase_log_map(quat_to_axis_angle(quats_xyzw))
    data = np.concatenate([phi, trans], 1)
    return spline_fit(stamps, data, order, knot_frequency)


def planar_points(width=1.5, height=1.5, delta=0.3):
    """test_utils.h:76-83."""
    nx, ny = int(width / delta) + 1, int(height / delta) + 1
    return np.array([[i * delta - width / 2, j * delta - height / 2, 0.0] for i in range(nx) for j in range(ny)])


def aprilgrid_points(rows=6, cols=6, tag_size=0.088, tag_spacing=0.3):
    """AprilGrid corner layout (aprilgrid_detector.cpp:28-50): feature id = 4*tag + corner."""
    pitch = tag_size * (1.0 + tag_spacing)
    pts = []
    for r in range(rows):
        for c in range(cols):
            x0, y0 = c * pitch, r * pitch
            for dx, dy in [(0, 0), (tag_size, 0), (tag_size, tag_size), (0, tag_size)]:
                pts.append([x0 + dx, y0 + dy, 0.0])
    pts = np.array(pts)
    pts[:, :2] -= pts[:, :2].mean(0)
    return pts


# ----------------------------------------------------------------------------
# sensor models (vectorised ProjectPoint / Project)
# ----------------------------------------------------------------------------
def project_point(model, k, p):
    """camera_models.h ProjectPoint x7; returns pixels (n,2), valid (n,)."""
    p = np.asarray(p, float)
    X, Y, Z = p[:, 0], p[:, 1], p[:, 2]
    f, cx, cy = k[0], k[1], k[2]
    with np.errstate(all="ignore"):
        if model in (1, 2):
            valid = Z > 0
            x, y = X / Z, Y / Z
            r2 = x * x + y * y
            k1, k2, p1, p2, k3 = k[3], k[4], k[5], k[6], k[7]
            s = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
            if model == 2:
                s = s / (1.0 + r2 * (k[8] + r2 * (k[9] + r2 * k[10])))
            px = x * s + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
            py = y * s + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y)
        elif model == 3:
            valid = Z > 0
            x, y = X / Z, Y / Z
            r = np.sqrt(x * x + y * y)
            th = np.arctan(r)
            th2 = th * th
            thd = th * (1.0 + th2 * (k[3] + th2 * (k[4] + th2 * (k[5] + th2 * k[6]))))
            r2 = r * r
            s = np.where(r < 1e-9, 1.0 + r2 * (k[3] - 1.0 / 3.0 + r2 * (-k[3] + k[4] + 0.2)),
                         thd / np.where(r < 1e-9, 1.0, r))
            px, py = x * s, y * s
        elif model == 4:
            xi, al = k[3], k[4]
            w1 = (1.0 - al) / al if al > 0.5 else al / (1.0 - al)
            w2sq = (w1 + xi) ** 2 / (2.0 * w1 * xi + xi * xi + 1.0)
            r2 = X * X + Y * Y + Z * Z
            valid = ~(Z * Z <= -w2sq * r2)
            r = np.sqrt(r2)
            d = np.sqrt(r2 * (1.0 + xi * xi) + 2.0 * xi * r * Z)
            s = 1.0 / (al * d + (1.0 - al) * (xi * r + Z))
            px, py = X * s, Y * s
        elif model == 5:
            w = k[3]
            valid = Z > 0
            x, y = X / Z, Y / Z
            r = np.sqrt(x * x + y * y)
            if w * w < 1e-5:
                s = np.ones_like(r)
            else:
                tt = 2.0 * np.tan(w * 0.5)
                s = np.where(r * r < 1e-5, tt / w, np.arctan(r * tt) / (np.where(r > 0, r, 1.0) * w))
            px, py = x * s, y * s
        elif model in (6, 7):
            al = k[3]
            w = (1.0 - al) / al if al > 0.5 else al / (1.0 - al)
            if model == 6:
                d = np.sqrt(X * X + Y * Y + Z * Z)
            else:
                d = np.sqrt(k[4] * np.sqrt(X * X + Y * Y) + Z * Z)
            valid = ~(Z <= -w * d)
            s = 1.0 / (al * d + (1.0 - al) * Z)
            px, py = X * s, Y * s
        else:
            raise ValueError("unknown camera model %r" % model)
    return np.stack([px * f + cx, py * f + cy], 1), valid


def imu_project(model, k, w):
    w = np.asarray(w, float)
    if model == 1:
        return k[0] * w
    if model == 2:
        return k[0] * w + np.asarray(k[1:4])
    if model == 3:
        sx, sy, sz, a1, a2, a3, a4, a5, a6, bx, by, bz = k
        return np.stack([bx + sx * (w[:, 0] + a1 * w[:, 1] + a2 * w[:, 2]),
                         by + sy * (w[:, 1] + a3 * w[:, 0] + a4 * w[:, 2]),
                         bz + sz * (w[:, 2] + a5 * w[:, 0] + a6 * w[:, 1])], 1)
    raise ValueError("unknown imu model %r" % model)


def project_camera(spl, model, intr, q_rc, t_rc, latency, times, points, q_wm, t_wm):
    """Camera::Project (camera.cpp:155-208) for one rigid body.

    Returns pixels (T*P,2), valid (T*P,), stamps (T*P,) [= t + latency, Q10],
    frame index and point index per row."""
    knots, basis, ctrl, order = spl
    pose = spline_eval(knots, basis, ctrl, order, times, 0)
    q_wr = quat_from_axis_angle(pose[:, :3])
    t_wr = pose[:, 3:]
    q_wc = quat_mul(q_wr, np.broadcast_to(q_rc, q_wr.shape))
    t_wc = quat_rotate(q_wr, t_rc) + t_wr
    q_cw = quat_conj(q_wc)
    t_cw = -quat_rotate(q_cw, t_wc)
    q_cb = quat_mul(q_cw, np.broadcast_to(q_wm, q_cw.shape))
    t_cb = quat_rotate(q_cw, t_wm) + t_cw
    T, P = len(times), len(points)
    pc = quat_rotate(q_cb[:, None, :], points[None, :, :]) + t_cb[:, None, :]
    pc = pc.reshape(T * P, 3)
    px, valid = project_point(model, intr, pc)
    valid &= pc[:, 2] > 0
    stamps = np.repeat(np.asarray(times, float) + latency, P)
    frame = np.repeat(np.arange(T), P)
    pidx = np.tile(np.arange(P), T)
    return px, valid, stamps, frame, pidx


def _imu_kinematics(spl, times):
    knots, basis, ctrl, order = spl
    seg = spline_index(knots, order, times)
    p = spline_eval(knots, basis, ctrl, order, times, 0, seg)
    pd = spline_eval(knots, basis, ctrl, order, times, 1, seg)
    pdd = spline_eval(knots, basis, ctrl, order, times, 2, seg)
    phi, phid, phidd = -p[:, :3], -pd[:, :3], -pdd[:, :3]
    J = exp_so3_jacobian(phi)
    H = exp_so3_hessian(phi)
    Jdot = np.einsum("nijk,nk->nji", H, phid)  # Jdot[:, :, i] = H[i] @ phid
    omega = np.einsum("nij,nj->ni", J, phid)
    alpha = np.einsum("nij,nj->ni", Jdot, phid) + np.einsum("nij,nj->ni", J, phidd)
    return phi, omega, alpha, pdd[:, 3:]


def project_gyroscope(spl, model, intr, q_rg, latency, times):
    """Gyroscope::Project (gyroscope.cpp:56-82)."""
    times = np.asarray(times, float)
    _, omega, _, _ = _imu_kinematics(spl, times)
    og = -quat_rotate(quat_conj(q_rg), omega)
    return imu_project(model, intr, og), times + latency


def project_accelerometer(spl, model, intr, q_ra, t_ra, latency, gravity, times):
    """Accelerometer::Project (accelerometer.cpp:76-123)."""
    times = np.asarray(times, float)
    phi, omega, alpha, acc = _imu_kinematics(spl, times)
    q_rw = quat_from_axis_angle(phi)
    t = np.asarray(t_ra, float)
    lever = np.cross(omega, np.cross(omega, t)) - np.cross(alpha, t)  # (ΩΩ + Α) t with Ω=-[ω]x, Α=-[α]x
    f = quat_rotate(quat_conj(q_ra), quat_rotate(q_rw, acc - np.asarray(gravity)) + lever)
    return imu_project(model, intr, f), times + latency


# ----------------------------------------------------------------------------
# scene description + flattening through the C ABI
# ----------------------------------------------------------------------------
@dataclass
class SensorSpec:
    kind: int
    model: int
    name: str
    intrinsics: np.ndarray
    q: np.ndarray            # x,y,z,w  (T_sensorrig_sensor rotation)
    t: np.ndarray
    latency: float
    intrinsics_true: np.ndarray
    q_true: np.ndarray
    t_true: np.ndarray
    latency_true: float
    enable_intrinsics: bool = True
    enable_extrinsics: bool = False
    enable_latency: bool = False
    sigma: float = 1.0
    loss: int = 0
    loss_scale: float = 1.0
    meas: Optional[np.ndarray] = None
    stamps: Optional[np.ndarray] = None
    point_idx: Optional[np.ndarray] = None
    is_outlier: Optional[np.ndarray] = None   # ground-truth gross-outlier flags (synthetic)

    @property
    def dim(self):
        return 2 if self.kind == _capi.SENSOR_CAMERA else 3

    @property
    def n(self):
        return 0 if self.stamps is None else len(self.stamps)


@dataclass
class Scene:
    order: int
    knots: np.ndarray
    basis: np.ndarray
    ctrl: np.ndarray
    ctrl_true: np.ndarray
    points: np.ndarray
    body_q: np.ndarray
    body_t: np.ndarray
    gravity: np.ndarray
    sensors: List[SensorSpec] = field(default_factory=list)
    body_pose_constant: bool = True
    points_constant: object = True   # bool, or one flag per model point

    @property
    def num_blocks(self):
        return sum(s.n for s in self.sensors)


@dataclass
class BuiltProblem:
    problem: "_capi.Problem"
    ctrl_blocks: np.ndarray
    point_blocks: np.ndarray
    body_q_block: int
    body_t_block: int
    gravity_block: int
    sensor_ids: List[int]
    sensor_blocks: List[dict]


def build_problem(api, scene, device=0, obs_slices=None):
    """Flatten a Scene through the C ABI, in BatchOptimizer::Optimize's order
    (batch_optimizer.cpp:57-70): world model, trajectory, then per sensor its
    parameters and residuals. `obs_slices[i]` optionally restricts sensor i to
    a slice of its observations (multi-GPU sharding)."""
    P = _capi.Problem(api, device)
    # WorldModel::AddParametersToProblem (world_model.cpp:40-77)
    pc = np.broadcast_to(np.asarray(scene.points_constant, bool), (len(scene.points),))
    point_blocks = P.add_param_blocks(np.asarray(scene.points, float), constant=pc)
    bt = P.add_param_block(scene.body_t, constant=scene.body_pose_constant)
    bq = P.add_param_block(scene.body_q, _capi.MANIFOLD_EIGEN_QUATERNION, scene.body_pose_constant)
    grav = P.add_param_block(scene.gravity, constant=True)  # Q6: gravity can never be enabled
    body = P.add_rigid_body(bq, bt)
    # Trajectory::AddParametersToProblem (bspline.hpp:10-17)
    ctrl_blocks = P.add_param_blocks(np.asarray(scene.ctrl, float))
    P.set_spline(scene.order, scene.knots, scene.basis, ctrl_blocks)
    sids, sblocks = [], []
    for i, s in enumerate(scene.sensors):
        # Sensor::AddParametersToProblem (camera.cpp:92-113)
        bi = P.add_param_block(s.intrinsics, constant=not s.enable_intrinsics)
        bT = P.add_param_block(s.t, constant=not s.enable_extrinsics)
        bQ = P.add_param_block(s.q, _capi.MANIFOLD_EIGEN_QUATERNION, not s.enable_extrinsics)
        bl = P.add_param_block([s.latency], constant=not s.enable_latency)
        sid = P.add_sensor(s.kind, s.model, bi, bQ, bT, bl,
                           grav if s.kind == _capi.SENSOR_ACCELEROMETER else -1, s.sigma, s.loss, s.loss_scale)
        sl = slice(None) if obs_slices is None else obs_slices[i]
        if s.n:
            if s.kind == _capi.SENSOR_CAMERA:
                P.add_camera_residuals(sid, s.meas[sl], s.stamps[sl], np.full(len(s.stamps[sl]), body, np.int32),
                                       point_blocks[s.point_idx[sl]])
            else:
                P.add_imu_residuals(sid, s.meas[sl], s.stamps[sl])
        sids.append(sid)
        sblocks.append(dict(intrinsics=bi, t=bT, q=bQ, latency=bl))
    return BuiltProblem(P, ctrl_blocks, point_blocks, bq, bt, grav, sids, sblocks)


def read_back(built, scene):
    """Current estimates of every sensor block and the control points."""
    P = built.problem
    out = []
    for s, b in zip(scene.sensors, built.sensor_blocks):
        out.append(dict(intrinsics=P.get_param_block(b["intrinsics"], len(s.intrinsics)),
                        t=P.get_param_block(b["t"], 3), q=P.get_param_block(b["q"], 4),
                        latency=P.get_param_block(b["latency"], 1)[0]))
    ctrl = P.get_param_blocks(np.asarray(built.ctrl_blocks, np.int32), 6).reshape(-1, 6)
    return out, ctrl


# ----------------------------------------------------------------------------
# scene generators
# ----------------------------------------------------------------------------
_OPENCV5_TRUE = np.array([785, 640, 400, -3.149e-1, 1.069e-1, 1.616e-4, 1.141e-4, -1.853e-2])  # batch_optimizer_test.cpp:56-57
_KB_TRUE = np.array([785.0, 640, 400, -1.3e-2, 2.1e-3, -8.0e-4, 1.0e-4])
_TRUE_INTRINSICS = {
    1: _OPENCV5_TRUE,
    2: np.concatenate([_OPENCV5_TRUE, [1.0e-3, -2.0e-4, 1.0e-5]]),
    3: _KB_TRUE,
    4: np.array([400.0, 640, 400, -0.2, 0.55]),
    5: np.array([600.0, 640, 400, 0.9]),
    6: np.array([500.0, 640, 400, 0.55]),
    7: np.array([500.0, 640, 400, 0.55, 1.1]),
}
_IMU_TRUE = {
    1: np.array([1.3]),
    2: np.array([1.3, 0.01, -0.01, 0.01]),  # batch_optimizer_test.cpp:88,92
    3: np.array([1.01, 0.99, 1.02, 1e-3, -2e-3, 1.5e-3, -1e-3, 2e-3, 1e-3, 0.01, -0.01, 0.01]),
}


def _initial_intrinsics(kind, model, truth):
    """Perturbation of batch_optimizer_test.cpp:125-127,151: 1.01x, camera distortion zeroed."""
    init = 1.01 * truth
    if kind == _capi.SENSOR_CAMERA:
        if model in (1, 2, 3):
            init[3:] = 0.0
        else:
            init[3:] = truth[3:]  # projection-shape parameters stay at truth (not a distortion series)
    elif model == 3:
        init[3:9] = 0.0
    return init


def make_scene(n_cameras=1, camera_model=1, imu=False, imu_model=2, duration=None, cam_rate=None, imu_rate=None,
               knot_frequency=10.0, order=6, chart="plane", pixel_noise=0.0, gyro_noise=0.0, accel_noise=0.0,
               seed=0xCA11C0, robust=False, outlier_fraction=0.0, perturb=True, estimate_spline_from_truth=True,
               max_cam_obs=None, segment_duration=0.75, repeats=1, free_chart_pose=False, free_points=False, n_imus=1):
    """Synthetic rig problem in the style of ToyStereoCameraAndImuCalibration
    (batch_optimizer_test.cpp:32-213): camera 0 is the rig frame with free
    intrinsics; further cameras also estimate extrinsics + latency; the IMU
    sensors estimate intrinsics, rotation (+ lever arm for the accelerometer)
    and latency. Measurements come from the fitted spline (the truth)."""
    rng = np.random.default_rng(seed)
    stamps, quats, trans = default_synthetic_poses(segment_duration=segment_duration, repeats=repeats)
    knots, basis, ctrl_true = fit_trajectory(stamps, quats, trans, knot_frequency, order)
    spl = (knots, basis, ctrl_true, order)
    t_end = stamps[-1] if duration is None else min(duration, stamps[-1])
    deg = order - 1
    last_valid = knots[len(knots) - deg - 1]
    points = planar_points() if chart == "plane" else aprilgrid_points()
    body_q = np.array([0.0, 0.0, 0.0, 1.0])
    body_t = np.zeros(3)
    gravity = np.array([0.0, 0.0, -9.80665])
    sensors = []
    max_lat = 0.02

    def rand_unit():
        v = rng.standard_normal(3)
        return v / np.linalg.norm(v)

    if cam_rate is None:
        cam_times = stamps[stamps <= t_end]
    else:
        cam_times = np.arange(0.0, t_end + 1e-12, 1.0 / cam_rate)
    cam_times = cam_times[cam_times + max_lat <= last_valid]
    for c in range(n_cameras):
        truth = _TRUE_INTRINSICS[camera_model].copy()
        if c == 0:
            q_true, t_true, lat_true = np.array([0.0, 0, 0, 1.0]), np.zeros(3), 0.0
        else:
            q_true = quat_from_axis_angle(rand_unit() * (2.0 * np.pi / 180.0))
            t_true = 0.05 * rng.uniform(-1, 1, 3)
            lat_true = 0.01 if n_cameras == 2 else 0.0025 * c  # stamps = t + latency must stay inside the knots
        px, valid, st, frame, pidx = project_camera(spl, camera_model, truth, q_true, t_true, lat_true, cam_times,
                                                    points, body_q, body_t)
        px, st, pidx = px[valid], st[valid], pidx[valid]
        if max_cam_obs is not None and len(st) > max_cam_obs:
            keep = np.sort(rng.choice(len(st), max_cam_obs, replace=False))
            px, st, pidx = px[keep], st[keep], pidx[keep]
        is_out = np.zeros(len(st), bool)
        if pixel_noise > 0:
            px = px + pixel_noise * rng.standard_normal(px.shape)
        if outlier_fraction > 0:
            is_out = rng.random(len(st)) < outlier_fraction
            mag = rng.uniform(5.0, 50.0, (len(st), 1)) * np.sign(rng.standard_normal((len(st), 2)))
            px = np.where(is_out[:, None], px + mag, px)
        init = _initial_intrinsics(_capi.SENSOR_CAMERA, camera_model, truth) if perturb else truth.copy()
        t_init = t_true + (0.01 * rng.uniform(-1, 1, 3) if (perturb and c > 0) else 0.0)
        sensors.append(SensorSpec(
            _capi.SENSOR_CAMERA, camera_model, "cam%d" % c, init, q_true.copy(), t_init,
            0.0 if perturb else lat_true, truth, q_true, t_true,
            lat_true, True, c > 0, c > 0, sigma=pixel_noise if pixel_noise > 0 else 1.0,
            loss=2 if robust else 0, loss_scale=1.0, meas=px, stamps=st, point_idx=pidx.astype(np.int32),
            is_outlier=is_out))
    if imu:
        if imu_rate is None:
            imu_times = stamps[stamps <= t_end]
        else:
            imu_times = np.arange(0.0, t_end + 1e-12, 1.0 / imu_rate)
        imu_times = imu_times[imu_times + max_lat <= last_valid]
        imu_specs = []
        for u in range(n_imus):      # one gyroscope + one accelerometer per IMU (each with its own mounting and intrinsics)
            sfx = "" if u == 0 else str(u)
            imu_specs += [(_capi.SENSOR_GYROSCOPE, "gyro" + sfx, gyro_noise), (_capi.SENSOR_ACCELEROMETER, "accel" + sfx, accel_noise)]
        for kind, nm, noise in imu_specs:
            truth = _IMU_TRUE[imu_model].copy()
            q_true = quat_from_axis_angle(rand_unit() * (2.0 * np.pi / 180.0))
            lat_true = 0.02
            if kind == _capi.SENSOR_GYROSCOPE:
                t_true = np.zeros(3)
                m, st = project_gyroscope(spl, imu_model, truth, q_true, lat_true, imu_times)
            else:
                t_true = 0.05 * rng.uniform(-1, 1, 3)
                m, st = project_accelerometer(spl, imu_model, truth, q_true, t_true, lat_true, gravity, imu_times)
            if noise > 0:
                m = m + noise * rng.standard_normal(m.shape)
            init = _initial_intrinsics(kind, imu_model, truth) if perturb else truth.copy()
            t_init = t_true + (0.05 * rng.uniform(-1, 1, 3) if (perturb and kind == _capi.SENSOR_ACCELEROMETER) else 0)
            sensors.append(SensorSpec(kind, imu_model, nm, init, q_true.copy(), t_init,
                                      0.0 if perturb else lat_true, truth, q_true, t_true,
                                      lat_true, True, True, True, sigma=noise if noise > 0 else 1.0,
                                      loss=1 if robust else 0, loss_scale=1.0, meas=m, stamps=st))
    points_constant = True
    points_init = points
    if free_points:
        # model_definition_is_constant = false (world_model.cpp:52-61). Three non-collinear anchor points stay fixed
        # so that the chart keeps its gauge; the others start a few millimetres off.
        p0 = points[0]
        i1 = int(np.argmax(np.linalg.norm(points - p0, axis=1)))
        d = (points[i1] - p0) / np.linalg.norm(points[i1] - p0)
        off = (points - p0) - np.outer((points - p0) @ d, d)
        i2 = int(np.argmax(np.linalg.norm(off, axis=1)))
        points_constant = np.zeros(len(points), bool)
        points_constant[[0, i1, i2]] = True
        points_init = points + np.where(points_constant[:, None], 0.0, 2e-3 * rng.standard_normal(points.shape))
    return Scene(order, knots, basis, ctrl_true.copy(), ctrl_true, points_init, body_q, body_t, gravity, sensors,
                 body_pose_constant=not free_chart_pose, points_constant=points_constant)


def config_scene(index, seed=None):
    """The five BASELINE.json configs as concretised in SURVEY.md §8(d)."""
    sd = (0xCA11C0 + index) if seed is None else seed
    if index == 0:   # plumbing: ~500 reprojection residual blocks
        return make_scene(1, 1, False, cam_rate=4.0, duration=1.0, chart="april", seed=sd, max_cam_obs=500,
                          pixel_noise=0.1, segment_duration=1.0 / 23.9)
    if index == 1:   # single pinhole camera, 20k blocks
        return make_scene(1, 1, False, cam_rate=20.0, duration=6.95, chart="april", seed=sd, pixel_noise=0.1,
                          segment_duration=6.95 / 23.9)
    if index == 2:   # stereo KB + IMU, ~50k blocks
        return make_scene(2, 3, True, 2, cam_rate=20.0, imu_rate=200.0, duration=8.0, chart="april", seed=sd,
                          pixel_noise=0.1, gyro_noise=1.7e-4 * np.sqrt(200.0), accel_noise=2e-3 * np.sqrt(200.0),
                          segment_duration=8.0 / 23.9)
    if index == 3:   # north star: 4 cam + IMU, 100k camera blocks, robust kernels
        return make_scene(4, 1, True, 3, cam_rate=20.0, imu_rate=200.0, duration=8.7, chart="april", seed=sd,
                          pixel_noise=0.1, gyro_noise=1.7e-4 * np.sqrt(200.0), accel_noise=2e-3 * np.sqrt(200.0),
                          robust=True, segment_duration=8.7 / 23.9)
    if index == 4:   # 8 cam + 2 IMU, 500k blocks, 2% gross outliers (to be tagged: SURVEY 8(d) row 5)
        return make_scene(8, 1, True, 3, cam_rate=20.0, imu_rate=200.0, duration=21.7, chart="april", seed=sd,
                          pixel_noise=0.1, gyro_noise=1.7e-4 * np.sqrt(200.0), accel_noise=2e-3 * np.sqrt(200.0),
                          robust=True, outlier_fraction=0.02, repeats=2, segment_duration=21.7 / 47.9, n_imus=2)
    if index == 5:   # the north star at 50 Hz knots ("also report"): ~438 control points
        return make_scene(4, 1, True, 3, cam_rate=20.0, imu_rate=200.0, duration=8.7, chart="april", seed=0xCA11C0 + 3,
                          pixel_noise=0.1, gyro_noise=1.7e-4 * np.sqrt(200.0), accel_noise=2e-3 * np.sqrt(200.0),
                          robust=True, segment_duration=8.7 / 23.9, knot_frequency=50.0)
    if index == 6:   # the shape of the one run the reference publishes (demos/imu_camera_calibration.ipynb: EuRoC,
        # one KannalaBrandt camera ~184k blocks, VectorNav gyro + accel ~14.4k each, ~1445 control points at 10 Hz knots)
        return make_scene(1, 3, True, 3, cam_rate=8.87, imu_rate=100.0, duration=144.2, chart="april", seed=sd,
                          pixel_noise=0.1, gyro_noise=1.7e-4 * np.sqrt(100.0), accel_noise=2e-3 * np.sqrt(100.0),
                          robust=True, repeats=6, segment_duration=144.2 / (6 * 23.9))
    raise ValueError(index)
