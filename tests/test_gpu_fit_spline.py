"""Spline initialisation on the device (`-m gpu`): calico_fit_spline against the reference's own known answers
(bspline_test.cpp:19-31, 52-94: fit of (cos t, sin 1.5 t, t cos t) at 10 Hz with 5 Hz knots; derivative tolerances
1e-6 / 1e-5 / 1e-4 / 1e-2) and against the oracle's FitToData on the same samples."""
import ctypes as C

import numpy as np
import pytest

import helpers
from calico_amd import _capi, synthetic as syn

pytestmark = pytest.mark.gpu


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fit_on_device(hip, t, data, order, knot_frequency):
    knots = syn.knot_vector(t[0], t[-1], order, knot_frequency)
    basis = np.ascontiguousarray(syn.basis_matrices(knots, order))
    ctrl = np.zeros((len(knots) - order, 6))
    rc = hip.fit_spline(0, order, len(knots), dp(knots), dp(basis), len(t), dp(t), dp(np.ascontiguousarray(data)), dp(ctrl))
    return rc, knots, basis, ctrl


def test_fit_reproduces_reference_known_answers(hip):
    t = 0.1 * np.arange(101)
    data = np.zeros((101, 6))
    data[:, 0], data[:, 1], data[:, 2] = np.cos(t), np.sin(1.5 * t), t * np.cos(t)
    rc, knots, basis, ctrl = _fit_on_device(hip, t, data, 6, 5.0)
    assert rc == 0
    ti = (t[-1] - t[0]) / 201 * np.arange(201)
    expect = [
        np.stack([np.cos(ti), np.sin(1.5 * ti), ti * np.cos(ti)], 1),
        np.stack([-np.sin(ti), 1.5 * np.cos(1.5 * ti), np.cos(ti) - ti * np.sin(ti)], 1),
        np.stack([-np.cos(ti), -2.25 * np.sin(1.5 * ti), -2.0 * np.sin(ti) - ti * np.cos(ti)], 1),
        np.stack([np.sin(ti), -3.375 * np.cos(1.5 * ti), ti * np.sin(ti) - 3.0 * np.cos(ti)], 1),
    ]
    for d, tol in enumerate([1e-6, 1e-5, 1e-4, 1e-2]):       # bspline_test.cpp:52-94
        out = syn.spline_eval(knots, basis, ctrl, 6, ti, d)
        assert np.abs(out[:, :3] - expect[d]).max() < tol


@pytest.mark.parametrize("order", [4, 6])
def test_fit_matches_oracle(order, hip, oracle):
    """Well-posed fit (more samples than control points everywhere): same control points as the oracle's FitToData
    (QR on the design matrix) to 1e-8 relative (normal equations square the conditioning of the B-spline basis)."""
    rng = np.random.default_rng(5)
    t = np.sort(rng.uniform(0.0, 12.0, 2000))
    t[0], t[-1] = 0.0, 12.0
    data = np.stack([np.sin(0.7 * t + i) * (1.0 + 0.1 * i) + 0.05 * t for i in range(6)], 1)
    rc, knots, basis, ctrl = _fit_on_device(hip, t, data, order, 10.0)
    assert rc == 0
    lib = oracle.lib
    lib.oracle_spline_create.restype = C.c_void_p
    s = C.c_void_p(lib.oracle_spline_create())
    lib.oracle_spline_fit_vectors.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int32]
    assert lib.oracle_spline_fit_vectors(s, len(t), dp(t), dp(np.ascontiguousarray(data)), C.c_double(10.0), order) == 0
    ref = np.zeros_like(ctrl)
    lib.oracle_spline_get.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oracle_spline_get(s, None, None, dp(ref))
    assert np.abs(ctrl - ref).max() <= 1e-8 * np.abs(ref).max()


def test_fit_error_conventions(hip):
    t = 0.1 * np.arange(50)
    data = np.zeros((50, 6))
    knots = syn.knot_vector(t[0], t[-1], 6, 5.0)
    basis = np.ascontiguousarray(syn.basis_matrices(knots, 6))
    ctrl = np.zeros((len(knots) - 6, 6))
    bad = t.copy(); bad[10] = bad[9] - 1.0                      # unsorted
    assert hip.fit_spline(0, 6, len(knots), dp(knots), dp(basis), 50, dp(bad), dp(data), dp(ctrl)) == 3
    far = t.copy(); far[-1] = 100.0                             # outside the valid knots
    assert hip.fit_spline(0, 6, len(knots), dp(knots), dp(basis), 50, dp(far), dp(data), dp(ctrl)) == 3
    assert hip.fit_spline(0, 1, len(knots), dp(knots), dp(basis), 50, dp(t), dp(data), dp(ctrl)) == 3   # order < 2
