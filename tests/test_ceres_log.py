"""The trust-region control pinned to the one per-iteration output of ceres::Solve the reference ships: the iteration
table of demos/imu_camera_calibration.ipynb (tests/golden/ceres_log_imu_camera.npz, extracted by
tests/golden/make_ceres_log.py). The table holds (tr_ratio, tr_radius) pairs for accepted steps, rejected steps and
candidates whose cost could not be evaluated (cost 1.797693e+308); replaying its tr_ratio column through the
oracle's control (CPU) and through the device's control kernel (`-m gpu`) must reproduce the tr_radius column, the
/2 /4 /8 /16 rejection ladder, and what the cost column shows on a rejected step."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ceres_log_imu_camera.npz")
DBL_MAX = np.finfo(np.float64).max


def _table():
    d = np.load(GOLD)
    rho = d["tr_ratio"][1:].copy()          # row 0 is the initial evaluation
    radius = d["tr_radius"]
    cost = d["cost"][1:]
    infinite = np.isinf(rho) | (rho < -1e300)
    rho[infinite] = 0.0
    return rho, infinite.astype(np.int32), radius, cost, d["cost_change"][1:]


def _check(radius_out, accepted, cost_col, rho, infinite, radius, cost):
    # every radius of the table to its three printed digits; rho is printed with three digits too, which moves an
    # accepted step's radius by up to ~0.6 % (d radius / d rho near rho = 0.9), and the replay carries that along
    np.testing.assert_allclose(radius_out, radius[1:], rtol=8e-3)
    assert list(accepted) == [int((not i) and r > 1e-3) for r, i in zip(rho, infinite)]
    # the rejection ladder is exact: consecutive rejections divide by 2, 4, 8, 16; the first one after an accepted step by 2
    prev, factor = radius[0], 2.0
    for k in range(len(rho)):
        if accepted[k]:
            factor = 2.0
        else:
            assert radius_out[k] == prev / factor, k
            factor *= 2.0
        prev = radius_out[k]
    assert radius_out[0] == 5e3 and radius_out[1] == 1.25e3 and radius_out[2] == 156.25 and radius_out[3] == 9.765625
    # cost column: a candidate that cannot be evaluated shows 1.797693e+308 (cost_change -1.80e+308), like the table
    for k in range(len(rho)):
        if infinite[k]:
            assert cost_col[k] == DBL_MAX and cost[k] == pytest.approx(1.797693e308, rel=1e-6)
    # a rejected step shows the CANDIDATE's cost (table rows 13, 14, 16, 17: cost above the previous row's, cost_change < 0)
    rejected = [k for k in range(len(rho)) if not accepted[k] and not infinite[k]]
    assert rejected == [12, 13, 15, 16]
    for k in rejected:
        assert cost_col[k] == pytest.approx(1.0 - rho[k], rel=1e-12) and cost_col[k] > 1.0
        last_accepted = max(j for j in range(k) if accepted[j])
        assert cost[k] > cost[last_accepted]      # the table's row shows a cost ABOVE the current point's: the candidate's


def test_golden_table_is_the_reference_table():
    d = np.load(GOLD)
    assert len(d["iteration"]) == 19 and list(d["iteration"]) == list(range(19))
    assert d["tr_radius"][0] == 1e4 and d["cost"][0] == pytest.approx(3.616876e11)
    # the four rows the reference's Ceres could not evaluate
    assert np.all(np.isinf(d["tr_ratio"][1:5])) and list(d["tr_radius"][1:5]) == [5e3, 1.25e3, 156.0, 9.77]


def test_oracle_control_reproduces_ceres_table():
    rho, infinite, radius, cost, _ = _table()
    lib = helpers.oracle_lib()
    n = len(rho)
    rad = np.zeros(n)
    acc = np.zeros(n, np.int32)
    col = np.zeros(n)
    fn = lib.oracle_lm_control_replay
    fn.restype = C.c_int32
    fn.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_double, C.c_double, C.c_double,
                   C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    o = helpers.oracle_api().default_options()
    assert o.initial_trust_region_radius == radius[0]        # batch_optimizer.cpp:10-17 leaves Ceres' default 1e4
    st = fn(n, rho.ctypes.data_as(C.POINTER(C.c_double)), infinite.ctypes.data_as(C.POINTER(C.c_int32)),
            o.initial_trust_region_radius, o.min_relative_decrease, o.max_trust_region_radius,
            rad.ctypes.data_as(C.POINTER(C.c_double)), acc.ctypes.data_as(C.POINTER(C.c_int32)),
            col.ctypes.data_as(C.POINTER(C.c_double)))
    assert st == 0
    _check(rad, acc, col, rho, infinite, radius, cost)


@pytest.mark.gpu
def test_device_control_reproduces_ceres_table():
    rho, infinite, radius, cost, _ = _table()
    api = helpers.hip_api()
    n = len(rho)
    rad = np.zeros(n)
    acc = np.zeros(n, np.int32)
    col = np.zeros(n)
    o = api.default_options()
    st = api.debug_lm_control_replay(0, n, rho.ctypes.data_as(C.POINTER(C.c_double)),
                                     infinite.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(o),
                                     rad.ctypes.data_as(C.POINTER(C.c_double)), acc.ctypes.data_as(C.POINTER(C.c_int32)),
                                     col.ctypes.data_as(C.POINTER(C.c_double)))
    assert st == 0
    _check(rad, acc, col, rho, infinite, radius, cost)
