# three_frames_per_cell with other IMU rates / seeds: worst estimate difference in units of the test's bound, plan shape
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch; torch.zeros(1, device="cuda")
import helpers, test_gpu_parity as t
from calico_amd import synthetic as syn
hip, oracle = helpers.hip_api(), helpers.oracle_api()
for rate in (50.0, 100.0, 150.0):
    for seed in (19, 23):
        scene = syn.make_scene(2, 1, True, 3, cam_rate=30.0, imu_rate=rate, duration=3.0, segment_duration=3.0 / 23.9, pixel_noise=0.1,
                               gyro_noise=1e-3, accel_noise=1e-2, robust=True, seed=seed)
        gpu, ref, sg, sr = t.solve_both(scene, hip, oracle, max_iter=50)
        info = gpu.problem.plan_info()
        eg, cg = syn.read_back(gpu, scene); er, cr = syn.read_back(ref, scene)
        worst = 0.0
        for a, b in zip(eg, er):
            for key in ("intrinsics", "t", "q"):
                worst = max(worst, np.abs(a[key] - b[key]).max() / (1e-6 * max(1e-3, np.abs(b[key]).max())))
        print("imu %5.0f Hz seed %d: fuse_expand %d frames/cell %d items/cell %d  iterations %d / %d  term %d/%d  worst %.3f of the bound" % (
            rate, seed, info["fuse_expand"], info["max_frames_per_cell"], info["max_items_per_cell"], sg.num_iterations, sr.num_iterations,
            sg.termination_type, sr.termination_type, worst), flush=True)
