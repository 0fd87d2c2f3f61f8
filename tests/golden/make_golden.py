#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/: small seeded problems (inputs) together with
the CPU oracle's outputs on them (cost, gradient, JtJ, residuals, inlier mask, converged
parameters). The reference itself cannot be built or imported in the authoring container (Eigen,
Ceres, abseil absent - SURVEY.md §8c), so these vectors are produced by the oracle, which is pinned
against the reference tests' known answers in tests/test_oracle_*.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from calico_amd import synthetic as syn  # noqa: E402
from golden_io import scene_to_dict  # noqa: E402

CASES = {
    "opencv5_stereo_imu_sb": dict(n_cameras=2, camera_model=1, imu=True, imu_model=2, robust=False),
    "kb_mono_imu_vn_robust": dict(n_cameras=1, camera_model=3, imu=True, imu_model=2, robust=True, outlier_fraction=0.03),
    "double_sphere_mono": dict(n_cameras=2, camera_model=4, imu=False),
    "eucm_free_chart": dict(n_cameras=2, camera_model=7, imu=False, free_chart_pose=True),
    "opencv5_free_model_points": dict(n_cameras=2, camera_model=1, imu=True, imu_model=2, free_points=True),
}


def main():
    api = helpers.oracle_api()
    only = sys.argv[1:]
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        scene = syn.make_scene(cam_rate=5.0, imu_rate=25.0, duration=2.0, segment_duration=2.0 / 23.9, pixel_noise=0.1,
                               gyro_noise=1e-3, accel_noise=1e-2, seed=1234, max_cam_obs=400, **kw)
        built = syn.build_problem(api, scene)
        cost, g, H = built.problem.evaluate()
        out = scene_to_dict(scene)
        out.update(cost=cost, gradient=g, jtj=H)
        for i, s in enumerate(scene.sensors):
            r, v = built.problem.residuals(built.sensor_ids[i], s.n, s.dim)
            out["res%d" % i] = r
            out["mask%d" % i] = built.problem.inlier_mask(built.sensor_ids[i], s.n, 3.0)
        o = api.default_options()
        o.minimizer_progress_to_stdout = 0
        o.max_num_iterations = 100
        sm = built.problem.solve(o)
        assert sm.termination_type == 0, (name, sm.message)  # fixtures must be converged solves
        est, ctrl = syn.read_back(built, scene)
        out.update(final_cost=sm.final_cost, termination_type=sm.termination_type, num_iterations=sm.num_iterations,
                   ctrl_final=ctrl,
                   points_final=np.stack([built.problem.get_param_block(int(b), 3) for b in built.point_blocks]))
        for i, e in enumerate(est):
            out["intr_final%d" % i] = e["intrinsics"]
            out["q_final%d" % i] = e["q"]
            out["t_final%d" % i] = e["t"]
            out["lat_final%d" % i] = np.array([e["latency"]])
            out["mask_final%d" % i] = built.problem.inlier_mask(built.sensor_ids[i], scene.sensors[i].n, 3.0)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "blocks", scene.num_blocks, "cost %.6e -> %.6e in %d its" % (cost, sm.final_cost, sm.num_iterations),
              "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
