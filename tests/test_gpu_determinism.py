"""Run-to-run determinism of the HIP path (`-m gpu`): no atomics, fixed-order reductions, so the same solve
repeated must reproduce every iterate bit for bit. A data race between waves or kernels shows up here first."""
import numpy as np
import pytest

import helpers
from calico_amd import _capi, synthetic as syn


def _solve_repeatedly(api, scene, repeats, max_iter):
    built = syn.build_problem(api, scene, device=0)
    P = built.problem
    init = [(int(b), scene.ctrl[i].copy()) for i, b in enumerate(built.ctrl_blocks)]
    for s, sb in zip(scene.sensors, built.sensor_blocks):
        init += [(sb["intrinsics"], s.intrinsics.copy()), (sb["t"], s.t.copy()), (sb["q"], s.q.copy()),
                 (sb["latency"], np.array([s.latency]))]
    ids = np.array([b for b, _ in init], np.int32)
    sizes = [int(np.asarray(v).size) for _, v in init]
    vals = np.concatenate([np.asarray(v, float).ravel() for _, v in init])
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = max_iter
    runs = []
    for _ in range(repeats):
        P.set_param_blocks(ids, vals)
        s = P.solve(o)
        costs = tuple(float(i.cost) for i in P.iterations())
        x = np.concatenate([np.asarray(P.get_param_block(int(b), n), float).ravel() for b, n in zip(ids, sizes)])
        runs.append((s.num_iterations, s.termination_type, costs, x))
    return runs


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["toy_stereo_imu", "four_cam_imu_robust"])
def test_repeated_solves_are_bit_identical(name):
    api = helpers.hip_api()
    if name == "toy_stereo_imu":          # long trajectory: the band is split by a separator (two segments)
        scene = syn.make_scene(2, 1, True, 2, seed=4)
    else:                                  # camera frames + IMU items + robust kernels, short trajectory
        scene = syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=2.0, chart="april", seed=11,
                               pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=0.2)
    runs = _solve_repeatedly(api, scene, repeats=4, max_iter=15)
    ref = runs[0]
    assert ref[0] > 0
    for r in runs[1:]:
        assert r[0] == ref[0] and r[1] == ref[1]
        assert r[2] == ref[2], "per-iteration costs differ between identical solves"
        assert np.array_equal(r[3], ref[3]), "estimates differ between identical solves"
