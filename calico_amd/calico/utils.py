"""Host-side helpers with the names and return conventions of the reference's calico/utils.py (numpy only; the
reference leans on OpenCV for the homography and the image resize). Cheap CPU work either side of the optimiser:
they stay on the host by design."""
from typing import Dict, List, Tuple

import numpy as np

from .. import _calico


def ComputeRmseHeatmapAndFeatureCount(measurement_residual_pairs, image_width: int, image_height: int,
                                      num_rows: int = 8, num_cols: int = 12):
    """utils.py:12-50. Bins the residuals of GetMeasurementResidualPairs() over a num_rows x num_cols grid of the
    image. Returns (heatmap stretched to image_height x image_width by nearest neighbour, binned RMSE, counts);
    empty bins are NaN like the reference's 0/0."""
    n = len(measurement_residual_pairs)
    px = np.array([m.pixel for m, _ in measurement_residual_pairs], float).reshape(n, 2)
    sq = np.array([float(np.sum(np.asarray(r) ** 2)) for _, r in measurement_residual_pairs], float)
    col = np.clip(np.floor(px[:, 0] / image_width * num_cols).astype(int), 0, num_cols - 1)
    row = np.clip(np.floor(px[:, 1] / image_height * num_rows).astype(int), 0, num_rows - 1)
    count = np.zeros((num_rows, num_cols))
    total = np.zeros((num_rows, num_cols))
    np.add.at(count, (row, col), 1.0)
    np.add.at(total, (row, col), sq)
    with np.errstate(invalid="ignore", divide="ignore"):
        rmse = np.sqrt(total / count)
    # nearest-neighbour stretch: destination pixel centre -> source bin
    r_idx = np.minimum((np.arange(image_height) * num_rows) // image_height, num_rows - 1)
    c_idx = np.minimum((np.arange(image_width) * num_cols) // image_width, num_cols - 1)
    return rmse[np.ix_(r_idx, c_idx)], rmse, count


def DetectionsToCameraMeasurements(detections: Dict[int, np.ndarray], stamp: float, seq: int):
    """utils.py:81-99: one CameraMeasurement per detected feature; model_id is 0 (single chart)."""
    out = []
    for feature_id, point in detections.items():
        m = _calico.CameraMeasurement()
        m.id.stamp = stamp
        m.id.image_id = seq
        m.id.model_id = 0
        m.id.feature_id = int(feature_id)
        m.pixel = np.asarray(point, float)
        out.append(m)
    return out


def _normalising_transform(p):
    c = p.mean(axis=0)
    d = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
    s = np.sqrt(2.0) / d if d > 0 else 1.0
    return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])


def _homography(src, dst, refine_iterations=10):
    """dst ~ H [src; 1]: normalised DLT, then a few Gauss-Newton steps on the transfer error (what
    cv2.findHomography does after its linear estimate)."""
    Ts, Td = _normalising_transform(src), _normalising_transform(dst)
    s = (Ts @ np.c_[src, np.ones(len(src))].T).T
    d = (Td @ np.c_[dst, np.ones(len(dst))].T).T
    A = np.zeros((2 * len(src), 9))
    A[0::2, 0:3] = s
    A[0::2, 6:9] = -d[:, :1] * s
    A[1::2, 3:6] = s
    A[1::2, 6:9] = -d[:, 1:2] * s
    h = np.linalg.svd(A)[2][-1]
    H = np.linalg.inv(Td) @ h.reshape(3, 3) @ Ts
    H /= H[2, 2]
    sh = np.c_[src, np.ones(len(src))]
    for _ in range(refine_iterations):
        q = sh @ H.T
        w = q[:, 2]
        r = np.c_[q[:, 0] / w - dst[:, 0], q[:, 1] / w - dst[:, 1]].ravel()
        J = np.zeros((2 * len(src), 8))
        J[0::2, 0:3] = sh / w[:, None]
        J[0::2, 6:8] = -(q[:, 0] / w ** 2)[:, None] * sh[:, :2]
        J[1::2, 3:6] = sh / w[:, None]
        J[1::2, 6:8] = -(q[:, 1] / w ** 2)[:, None] * sh[:, :2]
        step = np.linalg.lstsq(J, -r, rcond=None)[0]
        H = (H.ravel() + np.r_[step, 0.0]).reshape(3, 3)
        if np.abs(step).max() < 1e-14:
            break
    return H


def InitializePinholeAndPoses(all_detections: List[Dict[int, np.ndarray]], model_definition: Dict[int, np.ndarray]
                              ) -> Tuple[list, List[np.ndarray], List[np.ndarray]]:
    """utils.py:102-186: Zhang's closed-form pinhole initialisation from planar-chart detections.
    Returns ([fx, fy, s, cx, cy], [R_chart_camera per frame], [t_chart_camera per frame])."""
    Hs = []
    V = np.zeros((2 * len(all_detections), 6))

    def v(H, i, j):   # Zhang (1998) eq. (7), columns i, j of H
        a, b = H[:, i], H[:, j]
        return np.array([a[0] * b[0], a[0] * b[1] + a[1] * b[0], a[1] * b[1],
                         a[2] * b[0] + a[0] * b[2], a[2] * b[1] + a[1] * b[2], a[2] * b[2]])
    for i, det in enumerate(all_detections):
        ids = list(det.keys())
        pix = np.array([det[k] for k in ids], float).reshape(-1, 2)
        mod = np.array([np.asarray(model_definition[k], float)[:2] for k in ids]).reshape(-1, 2)
        H = _homography(mod, pix)
        Hs.append(H)
        V[2 * i] = v(H, 0, 1)
        V[2 * i + 1] = v(H, 0, 0) - v(H, 1, 1)
    # b = [B11, B12, B22, B13, B23, B33] of B = K^-T K^-1 up to scale: null vector of V
    b = np.linalg.svd(V)[2][-1]
    if b[0] < 0:
        b = -b
    B11, B12, B22, B13, B23, B33 = b
    den = B11 * B22 - B12 ** 2
    v0 = (B12 * B13 - B11 * B23) / den
    lam = B33 - (B13 ** 2 + v0 * (B12 * B13 - B11 * B23)) / B11
    alpha = np.sqrt(lam / B11)
    beta = np.sqrt(lam * B11 / den)
    gamma = -B12 * alpha ** 2 * beta / lam
    u0 = gamma * v0 / beta - B13 * alpha ** 2 / lam
    intrinsics = [alpha, beta, gamma, u0, v0]
    K = np.array([[alpha, gamma, u0], [0, beta, v0], [0, 0, 1.0]])
    K_inv = np.linalg.inv(K)
    R_chart_camera, t_chart_camera = [], []
    for H in Hs:
        Rt = K_inv @ H
        scale = 0.5 * (np.linalg.norm(Rt[:, 0]) + np.linalg.norm(Rt[:, 1]))
        if Rt[2, 2] < 0:          # the chart is in front of the camera
            scale = -scale
        R = np.c_[Rt[:, 0] / scale, Rt[:, 1] / scale, np.cross(Rt[:, 0] / scale, Rt[:, 1] / scale)]
        U, _, Vt = np.linalg.svd(R)
        R = U @ Vt                # nearest rotation
        t = Rt[:, 2] / scale
        R_chart_camera.append(R.T)
        t_chart_camera.append(-R.T @ t)
    return intrinsics, R_chart_camera, t_chart_camera
