"""calico_get_residuals / calico_project are per sensor (Sensor::UpdateResiduals, camera.cpp:56-80); the library evaluates
all blocks once and serves the other sensors from that evaluation until something the residuals depend on changes."""
import numpy as np
import pytest

import helpers
from calico_amd import synthetic as syn


@pytest.mark.gpu
def test_residuals_follow_parameter_measurement_and_tag_changes():
    api = helpers.hip_api()
    oracle = helpers.oracle_api()
    scene = syn.make_scene(2, 1, True, 2, seed=4, pixel_noise=0.2)
    built = syn.build_problem(api, scene, device=0)
    ref = syn.build_problem(oracle, scene)
    P, Q = built.problem, ref.problem

    def both(sid, sp):
        a, va = P.residuals(sid, sp.n, sp.dim)
        b, vb = Q.residuals(ref.sensor_ids[built.sensor_ids.index(sid)], sp.n, sp.dim)
        return a, va, b, vb

    # every sensor, twice (the second pass of the first sensor and all later sensors are served from the one evaluation)
    first = {}
    for _ in range(2):
        for sid, sp in zip(built.sensor_ids, scene.sensors):
            a, va, b, vb = both(sid, sp)
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
            assert np.array_equal(va, vb)
            first[sid] = a.copy()
    # prediction and residuals do not share the cache
    sid0, sp0 = built.sensor_ids[0], scene.sensors[0]
    pred, _ = P.project(sid0, sp0.n, sp0.dim)
    again, _ = P.residuals(sid0, sp0.n, sp0.dim)
    assert np.array_equal(again, first[sid0]) and not np.allclose(pred, again)
    # a parameter value changes: every sensor that depends on it follows (here a control point: all of them)
    blk = int(built.ctrl_blocks[3])
    v0 = np.asarray(P.get_param_block(blk, 6), float).copy()
    for prob in (P, Q):
        prob.set_param_block(int(blk) if prob is P else int(ref.ctrl_blocks[3]), v0 + 1e-3)
    for sid, sp in zip(built.sensor_ids, scene.sensors):
        a, va, b, vb = both(sid, sp)
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
    moved, _ = P.residuals(sid0, sp0.n, sp0.dim)
    assert not np.array_equal(moved, first[sid0])
    # ... and back: the old values again, bit for bit
    P.set_param_block(blk, v0)
    back, _ = P.residuals(sid0, sp0.n, sp0.dim)
    assert np.array_equal(back, first[sid0])
    # a tag changes what is reported for the tagged observation (camera.cpp:121-124: it is left out)
    mask = np.zeros(sp0.n, np.uint8)
    mask[5] = 1
    P.set_outlier_mask(sid0, mask)
    tagged, vt = P.residuals(sid0, sp0.n, sp0.dim, check=False)
    assert vt[5] == 0 and np.all(tagged[5] == 0.0) and np.array_equal(np.delete(tagged, 5, 0), np.delete(first[sid0], 5, 0))
    P.set_outlier_mask(sid0, None)
    clear, vc = P.residuals(sid0, sp0.n, sp0.dim)
    assert vc[5] == 1 and np.array_equal(clear, first[sid0])
