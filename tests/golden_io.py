"""(De)serialisation of synthetic scenes for the golden fixtures (plain numpy arrays only)."""
import numpy as np

from calico_amd import synthetic as syn

_SENSOR_ARRAYS = ["intrinsics", "q", "t", "intrinsics_true", "q_true", "t_true", "meas", "stamps"]
_SENSOR_SCALARS = ["kind", "model", "latency", "latency_true", "enable_intrinsics", "enable_extrinsics", "enable_latency",
                   "sigma", "loss", "loss_scale"]


def scene_to_dict(scene):
    d = dict(order=scene.order, knots=scene.knots, basis=scene.basis, ctrl=scene.ctrl, ctrl_true=scene.ctrl_true,
             points=scene.points, body_q=scene.body_q, body_t=scene.body_t, gravity=scene.gravity,
             body_pose_constant=scene.body_pose_constant, n_sensors=len(scene.sensors),
             points_constant=np.broadcast_to(np.asarray(scene.points_constant, bool), (len(scene.points),)).copy())
    for i, s in enumerate(scene.sensors):
        for k in _SENSOR_ARRAYS:
            d["s%d_%s" % (i, k)] = np.asarray(getattr(s, k))
        d["s%d_scalars" % i] = np.array([float(getattr(s, k)) for k in _SENSOR_SCALARS])
        d["s%d_point_idx" % i] = s.point_idx if s.point_idx is not None else np.zeros(0, np.int32)
    return d


def scene_from_dict(d):
    sensors = []
    for i in range(int(d["n_sensors"])):
        sc = dict(zip(_SENSOR_SCALARS, d["s%d_scalars" % i]))
        pidx = d["s%d_point_idx" % i]
        sensors.append(syn.SensorSpec(
            int(sc["kind"]), int(sc["model"]), "s%d" % i, d["s%d_intrinsics" % i].copy(), d["s%d_q" % i].copy(),
            d["s%d_t" % i].copy(), float(sc["latency"]), d["s%d_intrinsics_true" % i], d["s%d_q_true" % i], d["s%d_t_true" % i],
            float(sc["latency_true"]), bool(sc["enable_intrinsics"]), bool(sc["enable_extrinsics"]), bool(sc["enable_latency"]),
            float(sc["sigma"]), int(sc["loss"]), float(sc["loss_scale"]), d["s%d_meas" % i], d["s%d_stamps" % i],
            pidx.astype(np.int32) if len(pidx) else None))
    return syn.Scene(int(d["order"]), d["knots"], d["basis"], d["ctrl"].copy(), d["ctrl_true"], d["points"], d["body_q"],
                     d["body_t"], d["gravity"], sensors, body_pose_constant=bool(d["body_pose_constant"]),
                     points_constant=d["points_constant"] if "points_constant" in d else True)
