"""ctypes binding of the C ABI declared in include/calico_hip.h.

`CApi(lib, prefix)` binds one shared library exporting that ABI under a symbol
prefix. The product uses `load_hip()` (libcalico_hip.so, prefix "calico_");
there is no CPU fallback: if the HIP library is missing or no GPU is usable the
call fails loudly. tests/ bind the CPU oracle through the same class with
prefix "oracle_" (see tests/helpers.py) - never the product.
"""
import ctypes as C
import os

import numpy as np

OK, INVALID_ARGUMENT, FAILED_PRECONDITION, UNIMPLEMENTED, INTERNAL = 0, 3, 9, 12, 13
MANIFOLD_EUCLIDEAN, MANIFOLD_EIGEN_QUATERNION = 0, 1
SENSOR_CAMERA, SENSOR_GYROSCOPE, SENSOR_ACCELEROMETER = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2


class SolverOptions(C.Structure):
    """calico_solver_options (ceres::Solver::Options fields the path uses)."""
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("num_threads", C.c_int32),
        ("minimizer_progress_to_stdout", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("sync_every", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
    ]


class Summary(C.Structure):
    """calico_summary (ceres::Solver::Summary fields the reference binds)."""
    _fields_ = [
        ("termination_type", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("num_iterations", C.c_int32),
        ("num_jacobian_evaluations", C.c_int32),
        ("num_cost_evaluations", C.c_int32),
        ("num_residual_blocks", C.c_int32),
        ("num_residuals", C.c_int32),
        ("num_parameter_blocks", C.c_int32),
        ("num_parameters", C.c_int32),
        ("num_effective_parameters", C.c_int32),
        ("num_residual_blocks_reduced", C.c_int32),
        ("num_residuals_reduced", C.c_int32),
        ("num_parameter_blocks_reduced", C.c_int32),
        ("num_parameters_reduced", C.c_int32),
        ("num_effective_parameters_reduced", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("total_time_in_seconds", C.c_double),
        ("solve_time_in_seconds", C.c_double),
        ("message", C.c_char * 256),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["message"] = d["message"].decode()
        return d


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("step_is_valid", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("reserved", C.c_int32),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class CalicoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("Error: %s (status %d)" % (message, code))
        self.code = code
        self.message = message


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# Every symbol include/calico_hip.h declares (without prefix).
ABI_SYMBOLS = [
    "problem_create", "problem_destroy", "last_error", "default_solver_options",
    "problem_add_param_block", "problem_add_param_blocks", "get_param_block", "set_param_block", "set_param_blocks", "get_param_blocks",
    "problem_set_spline", "problem_add_rigid_body", "problem_add_sensor",
    "problem_add_camera_residuals", "problem_add_imu_residuals", "solve",
    "get_iterations", "get_residuals", "get_inlier_mask",
    "num_effective_parameters", "evaluate", "problem_set_allreduce", "problem_set_shard",
    "problem_set_stream", "get_phase_time", "set_phase_timing", "project",
    "problem_set_outlier_mask", "mark_outliers", "fit_spline", "residual_heatmap",
    "comm_get_unique_id", "comm_init_rccl", "comm_info", "problem_finalize", "plan_cache_stats", "plan_cache_clear",
]
# Test hooks (calico_amd/csrc/calico_hip_testing.h): exported, not part of the drop-in surface.
TEST_SYMBOLS = ["debug_lm_control_replay", "debug_plan_info", "debug_roll_table"]


class CApi:
    """Functions of one shared library exporting the calico C ABI."""

    def __init__(self, lib, prefix, has_device=True):
        self.lib = lib
        self.prefix = prefix
        self.has_device = has_device
        g = self._get
        P, D, I = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
        if has_device:
            g("problem_create", C.c_int32, [C.POINTER(P), C.c_int32])
        else:
            g("problem_create", C.c_int32, [C.POINTER(P)])
        g("problem_destroy", None, [P])
        g("last_error", C.c_char_p, [P])
        g("default_solver_options", None, [C.POINTER(SolverOptions)])
        g("problem_add_param_block", C.c_int32, [P, D, C.c_int32, C.c_int32, C.c_int32, I])
        if has_device:
            g("problem_add_param_blocks", C.c_int32, [P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), D, I])
        g("get_param_block", C.c_int32, [P, C.c_int32, D])
        g("set_param_block", C.c_int32, [P, C.c_int32, D])
        g("set_param_blocks", C.c_int32, [P, C.c_int32, I, D])
        g("get_param_blocks", C.c_int32, [P, C.c_int32, I, D])
        g("problem_set_spline", C.c_int32, [P, C.c_int32, C.c_int32, D, D, I])
        g("problem_add_rigid_body", C.c_int32, [P, C.c_int32, C.c_int32, I])
        g("problem_add_sensor", C.c_int32,
          [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
           C.c_double, C.c_int32, C.c_double, I])
        g("problem_add_camera_residuals", C.c_int32, [P, C.c_int32, C.c_int64, D, D, I, I])
        g("problem_add_imu_residuals", C.c_int32, [P, C.c_int32, C.c_int64, D, D])
        g("solve", C.c_int32, [P, C.POINTER(SolverOptions), C.POINTER(Summary)])
        g("get_iterations", C.c_int32, [P, C.POINTER(Iteration), C.c_int32, I])
        g("get_residuals", C.c_int32, [P, C.c_int32, D, C.POINTER(C.c_uint8)])
        g("get_inlier_mask", C.c_int32, [P, C.c_int32, C.c_double, C.POINTER(C.c_uint8)])
        g("num_effective_parameters", C.c_int32, [P, I])
        g("evaluate", C.c_int32, [P, D, D, D])
        if has_device:
            g("problem_set_allreduce", C.c_int32, [P, ALLREDUCE_FN, C.c_void_p])
            g("problem_set_shard", C.c_int32, [P, C.c_int32, C.c_int32])
            g("problem_set_stream", C.c_int32, [P, C.c_void_p])
            g("get_phase_time", C.c_int32, [P, C.c_int32, D, C.POINTER(C.c_int64)])
            g("project", C.c_int32, [P, C.c_int32, D, C.POINTER(C.c_uint8)])
            g("problem_set_outlier_mask", C.c_int32, [P, C.c_int32, C.POINTER(C.c_uint8)])
            g("mark_outliers", C.c_int32, [P, C.c_int32, C.c_double, C.POINTER(C.c_int64)])
            g("residual_heatmap", C.c_int32, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, D, C.POINTER(C.c_int64)])
            g("fit_spline", C.c_int32, [C.c_int32, C.c_int32, C.c_int32, D, D, C.c_int64, D, D, D])
            g("set_phase_timing", C.c_int32, [P, C.c_int32])
            g("problem_finalize", C.c_int32, [P])
            g("comm_get_unique_id", C.c_int32, [C.POINTER(C.c_uint8)])
            g("comm_init_rccl", C.c_int32, [P, C.POINTER(C.c_uint8), C.c_int32, C.c_int32])
            g("comm_info", C.c_int32, [P, I, I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
            g("plan_cache_stats", C.c_int32, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
            g("plan_cache_clear", C.c_int32, [])
            g("debug_lm_control_replay", C.c_int32,
              [C.c_int32, C.c_int32, D, I, C.POINTER(SolverOptions), D, I, D])
            g("debug_plan_info", C.c_int32, [P, I, C.c_int32])
            if hasattr(self.lib, self.prefix + "debug_roll_table"):      # (a library of an older round, loaded for a same-box A/B, has none)
                g("debug_roll_table", C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_uint32)])

    def _get(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, name, fn)

    def default_options(self):
        o = SolverOptions()
        self.default_solver_options(C.byref(o))
        return o


def plan_cache_stats(api):
    """(hits, misses, plans held) of the library's plan cache."""
    h, m, e = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    api.plan_cache_stats(C.byref(h), C.byref(m), C.byref(e))
    return h.value, m.value, e.value


def comm_unique_id(api):
    """128-byte RCCL id drawn by rank 0, to be handed to every rank's Problem.comm_init_rccl."""
    buf = (C.c_uint8 * 128)()
    st = api.comm_get_unique_id(buf)
    if st != OK:
        raise CalicoError(st, "calico_comm_get_unique_id failed")
    return bytes(buf)


class Problem:
    """Thin object wrapper of a calico_problem handle."""

    def __init__(self, api, device=0):
        self.api = api
        self.h = C.c_void_p()
        if api.has_device:
            st = api.problem_create(C.byref(self.h), int(device))
        else:
            st = api.problem_create(C.byref(self.h))
        if st != OK:
            raise CalicoError(st, "calico_problem_create failed (no usable HIP device?)")
        self._keep = []
        self._sizes = {}      # block id -> ambient size, of the blocks added through this object

    def close(self):
        if self.h:
            self.api.problem_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise CalicoError(st, self.api.last_error(self.h).decode())

    def add_param_block(self, values, manifold=MANIFOLD_EUCLIDEAN, constant=False):
        v = _f64(values).ravel()
        out = C.c_int32(-1)
        self._check(self.api.problem_add_param_block(self.h, _dp(v), v.size, manifold, int(bool(constant)),
                                                     C.byref(out)))
        self._sizes[out.value] = v.size
        return out.value

    def add_param_blocks(self, values, manifold=MANIFOLD_EUCLIDEAN, constant=False):
        """n blocks of one size (rows of `values`); one ABI call where the library has the bulk form. Returns the ids."""
        v = _f64(values)
        if len(v) == 0:
            return np.zeros(0, np.int32)
        v = v.reshape(len(v), -1)
        n, size = v.shape
        const = np.ascontiguousarray(np.broadcast_to(np.asarray(constant, bool), (n,)), dtype=np.uint8)
        if not hasattr(self.api, "problem_add_param_blocks"):
            return np.array([self.add_param_block(v[i], manifold, bool(const[i])) for i in range(n)], np.int32)
        ids = np.zeros(n, np.int32)
        self._check(self.api.problem_add_param_blocks(self.h, n, size, manifold, const.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(v), _ip(ids)))
        for i in ids:
            self._sizes[int(i)] = size
        return ids

    def get_param_block(self, block_id, size):
        out = np.zeros(size)
        self._check(self.api.get_param_block(self.h, block_id, _dp(out)))
        return out

    def get_param_blocks(self, block_ids, sizes):
        """Values of several blocks, concatenated (`sizes`: the blocks' sizes, or one size for all)."""
        ids = _i32(block_ids)
        per = np.broadcast_to(np.asarray(sizes, np.int64), (ids.size,))
        # the C call writes every block's real size and takes no capacity: the caller's sizes must be the blocks' own
        for i, n in zip(ids, per):
            known = self._sizes.get(int(i))
            if known is not None and known != int(n):
                raise ValueError("block %d has %d values, not %d" % (int(i), known, int(n)))
        total = int(per.sum())
        out = np.zeros(total)
        self._check(self.api.get_param_blocks(self.h, ids.size, _ip(ids), _dp(out)))
        return out

    def set_param_block(self, block_id, values):
        v = _f64(values).ravel()
        self._check(self.api.set_param_block(self.h, block_id, _dp(v)))

    def set_param_blocks(self, block_ids, values_concat):
        ids, v = _i32(block_ids), _f64(values_concat)
        self._check(self.api.set_param_blocks(self.h, ids.size, _ip(ids), _dp(v)))

    def set_spline(self, order, knots, basis, ctrl_block_ids):
        k, b, c = _f64(knots), _f64(basis), _i32(ctrl_block_ids)
        assert c.size == k.size - order
        self._check(self.api.problem_set_spline(self.h, order, k.size, _dp(k), _dp(b), _ip(c)))

    def add_rigid_body(self, q_block, t_block):
        out = C.c_int32(-1)
        self._check(self.api.problem_add_rigid_body(self.h, q_block, t_block, C.byref(out)))
        return out.value

    def add_sensor(self, kind, model, intr, q, t, lat, grav=-1, sigma=1.0, loss=0, loss_scale=1.0):
        out = C.c_int32(-1)
        self._check(self.api.problem_add_sensor(self.h, kind, model, intr, q, t, lat, grav, float(sigma), loss,
                                                float(loss_scale), C.byref(out)))
        return out.value

    def add_camera_residuals(self, sensor, pixels, stamps, body_ids, point_blocks):
        px, st, b, p = _f64(pixels), _f64(stamps), _i32(body_ids), _i32(point_blocks)
        self._check(self.api.problem_add_camera_residuals(self.h, sensor, st.size, _dp(px), _dp(st), _ip(b), _ip(p)))

    def add_imu_residuals(self, sensor, meas, stamps):
        m, st = _f64(meas), _f64(stamps)
        self._check(self.api.problem_add_imu_residuals(self.h, sensor, st.size, _dp(m), _dp(st)))

    def solve(self, options=None):
        o = options if options is not None else self.api.default_options()
        s = Summary()
        self._check(self.api.solve(self.h, C.byref(o), C.byref(s)))
        return s

    def iterations(self, max_rows=4096):
        buf = (Iteration * max_rows)()
        n = C.c_int32(0)
        self._check(self.api.get_iterations(self.h, buf, max_rows, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def residuals(self, sensor, n, dim, check=True):
        out = np.zeros((n, dim))
        valid = np.zeros(n, dtype=np.uint8)
        st = self.api.get_residuals(self.h, sensor, _dp(out), valid.ctypes.data_as(C.POINTER(C.c_uint8)))
        if check:
            self._check(st)
        return out, valid

    def project(self, sensor, n, dim):
        """Sensor::Project for the registered observations: model prediction per observation (device only)."""
        out = np.zeros((n, dim))
        valid = np.zeros(n, dtype=np.uint8)
        self._check(self.api.project(self.h, sensor, _dp(out), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out, valid

    def set_outlier_mask(self, sensor, is_outlier=None):
        """MarkOutliersById / ClearOutliers (device only): one byte per observation, None clears."""
        if is_outlier is None:
            self._check(self.api.problem_set_outlier_mask(self.h, sensor, None))
        else:
            m = np.ascontiguousarray(is_outlier, dtype=np.uint8)
            self._check(self.api.problem_set_outlier_mask(self.h, sensor, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def mark_outliers(self, sensor, threshold):
        """One tagging pass on the device; returns the number of observations tagged by this call."""
        n = C.c_int64(0)
        self._check(self.api.mark_outliers(self.h, sensor, float(threshold), C.byref(n)))
        return n.value

    def residual_heatmap(self, sensor, image_width, image_height, num_rows=8, num_cols=12):
        """Binned RMSE and feature count of a camera's residuals (utils.py:12-50), reduced on the device."""
        rmse = np.zeros((num_rows, num_cols))
        count = np.zeros((num_rows, num_cols), dtype=np.int64)
        self._check(self.api.residual_heatmap(self.h, sensor, int(image_width), int(image_height), int(num_rows), int(num_cols),
                                              _dp(rmse), count.ctypes.data_as(C.POINTER(C.c_int64))))
        return rmse, count

    def inlier_mask(self, sensor, n, threshold):
        mask = np.zeros(n, dtype=np.uint8)
        self._check(self.api.get_inlier_mask(self.h, sensor, float(threshold),
                                             mask.ctypes.data_as(C.POINTER(C.c_uint8))))
        return mask

    def num_effective_parameters(self):
        n = C.c_int32(0)
        self._check(self.api.num_effective_parameters(self.h, C.byref(n)))
        return n.value

    def evaluate(self, want_jtj=True):
        n = self.num_effective_parameters()
        cost = C.c_double(0)
        g = np.zeros(n)
        H = np.zeros((n, n)) if want_jtj else None
        self._check(self.api.evaluate(self.h, C.byref(cost), _dp(g), _dp(H) if want_jtj else None))
        return cost.value, g, H

    def set_allreduce(self, pyfunc):
        cb = ALLREDUCE_FN(pyfunc)
        self._keep.append(cb)
        self._check(self.api.problem_set_allreduce(self.h, cb, None))

    def finalize(self):
        self._check(self.api.problem_finalize(self.h))

    def plan_info(self):
        """Test hook (calico_hip_testing.h): which evaluation route / solver the plan of this handle takes."""
        out = np.zeros(9, np.int32)
        self._check(self.api.debug_plan_info(self.h, out.ctypes.data_as(C.POINTER(C.c_int32)), 9))
        keys = ("fuse_expand", "frames", "items", "cells", "max_frames_per_cell", "max_items_per_cell", "tree_solver", "m",
                "all_control_points_observed")
        return dict(zip(keys, (int(v) for v in out)))

    def comm_init_rccl(self, unique_id, rank, world_size):
        """Native exchange: the handle creates its own RCCL communicator from the 128-byte id (see comm_unique_id)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.api.comm_init_rccl(self.h, buf, int(rank), int(world_size)))

    def comm_info(self):
        """(rank, world, local residual blocks, total residual blocks) as the handle's communicator / shard sees them."""
        r, w = C.c_int32(0), C.c_int32(0)
        nl, nt = C.c_int64(0), C.c_int64(0)
        self._check(self.api.comm_info(self.h, C.byref(r), C.byref(w), C.byref(nl), C.byref(nt)))
        return r.value, w.value, nl.value, nt.value

    def set_shard(self, rank, world_size):
        self._check(self.api.problem_set_shard(self.h, int(rank), int(world_size)))

    def set_stream(self, stream_ptr):
        self._check(self.api.problem_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_phase_timing(self, mask):
        self._check(self.api.set_phase_timing(self.h, int(mask)))

    def phase_time(self, phase):
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._check(self.api.get_phase_time(self.h, phase, C.byref(ms), C.byref(n)))
        return ms.value, n.value


_HIP_LIB_NAME = "libcalico_hip.so"
_hip_api = None


def hip_library_path():
    # CALICO_HIP_LIB (only with CALICO_DEV=1): another build of the same library, for development A/B runs
    # (profiles/dev/ab.sh); the product path is the in-tree one
    other = os.environ.get("CALICO_HIP_LIB") if os.environ.get("CALICO_DEV") == "1" else None
    return other or os.path.join(os.path.dirname(os.path.abspath(__file__)), _HIP_LIB_NAME)


def load_hip():
    """Load libcalico_hip.so (built in-tree by __graft_entry__.build()).

    Raises if it is missing - the product has no other backend."""
    global _hip_api
    if _hip_api is None:
        path = hip_library_path()
        if not os.path.exists(path):
            raise RuntimeError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP library is the only backend; there is no CPU fallback)" % path)
        _hip_api = CApi(C.CDLL(path), "calico_", has_device=True)
    return _hip_api
