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
    assert len(rho) == 150
    # Row by row: the radius CHANGE of every row to the table's printed digits. Radius and rho are printed with three
    # digits each (a rounding of up to 0.5 % per radius, and d radius / d rho of an accepted step moves it by up to
    # ~0.6 % more near rho = 0.9), so a row's ratio radius[k] / radius[k-1] is good to ~1 % (observed: 0.51 %).
    prev_out = np.concatenate([[radius[0]], radius_out[:-1]])
    np.testing.assert_allclose(radius_out / prev_out, radius[1:] / radius[:-1], rtol=8e-3)
    # The replay carries its own radius along all 150 rows; the roundings of the printed rho accumulate like a random
    # walk: the whole table stays within 1 % (observed: 0.82 % at row 31, 0.4 % at row 150)
    np.testing.assert_allclose(radius_out, radius[1:], rtol=1e-2)
    # the regimes the first 18 rows do not hold:
    #  * rho > 1 (rows 22-24: 1.04, 1.11, 1.09): 1 - (2 rho - 1)^3 < 1/3, the radius triples
    for k in (22, 23, 24):
        assert rho[k - 1] > 1.0 and radius_out[k - 1] == pytest.approx(3.0 * prev_out[k - 1], rel=1e-12)
    #  * an ACCEPTED step that shrinks the radius (row 35: rho = 0.256 -> 1 - (2 rho - 1)^3 = 1.116)
    assert accepted[34] and rho[34] == pytest.approx(0.256) and radius_out[34] < prev_out[34]
    assert radius_out[34] / prev_out[34] == pytest.approx(1.0 / (1.0 - (2 * 0.256 - 1) ** 3), rel=1e-12)
    #  * slow growth (rows 40-150: rho ~ 0.54-0.6 -> x 1.0005-1.008 per step; the table creeps from 3.78e6 to 4.29e6)
    tail = radius_out[39:] / prev_out[39:]
    assert np.all(tail > 1.0) and np.all(tail < 1.01) and np.all(accepted[39:] == 1)
    assert radius_out[-1] == pytest.approx(4.29e6, rel=1e-2)
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
    # all 151 rows of the notebook's table: its stdout is stored in two chunks (the cut falls inside row 19), stitched by
    # make_ceres_log.py; the run ends at row 150 = its max_num_iterations (NO_CONVERGENCE)
    assert len(d["iteration"]) == 151 and list(d["iteration"]) == list(range(151))
    assert d["cost"][19] == pytest.approx(2.914217e8) and d["tr_ratio"][19] == 0.966 and d["tr_radius"][19] == 26.9
    assert d["cost"][150] == pytest.approx(8.666563e6) and d["tr_radius"][150] == 4.29e6
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


def _budget_scene():
    from calico_amd import synthetic as syn
    return syn.make_scene(1, 1, True, 2, cam_rate=10.0, imu_rate=50.0, duration=2.0, segment_duration=2.0 / 23.9,
                          pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, max_cam_obs=600)


def _ends_like_the_table(api):
    """The table ends at row 150 = the notebook's max_num_iterations: the minimizer logs that row and stops with
    NO_CONVERGENCE ("Maximum number of iterations reached."), Summary::iterations holding max + 1 rows."""
    from calico_amd import _capi, synthetic as syn
    built = syn.build_problem(api, _budget_scene())
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 3
    o.function_tolerance = 0.0      # nothing else may end the solve first
    o.parameter_tolerance = 0.0
    o.gradient_tolerance = 0.0
    s = built.problem.solve(o)
    rows = built.problem.iterations()
    assert s.termination_type == _capi.NO_CONVERGENCE and s.message == b"Maximum number of iterations reached."
    assert s.num_iterations == 3 and [r.iteration for r in rows] == [0, 1, 2, 3]
    assert s.num_successful_steps + s.num_unsuccessful_steps == 3
    return s, rows


def test_oracle_stops_at_the_iteration_budget():
    _ends_like_the_table(helpers.oracle_api())


@pytest.mark.gpu
def test_device_stops_at_the_iteration_budget():
    s, rows = _ends_like_the_table(helpers.hip_api())
    so, rows_o = _ends_like_the_table(helpers.oracle_api())
    assert [r.step_is_successful for r in rows] == [r.step_is_successful for r in rows_o]
    for a, b in zip(rows, rows_o):
        assert a.cost == pytest.approx(b.cost, rel=1e-6) and a.trust_region_radius == pytest.approx(b.trust_region_radius, rel=1e-6)
