"""GPU parity: libcalico_hip.so (through the C ABI) against the CPU oracle on
identical seeded inputs. Integer outputs are compared bit-exact; floating point
within the tolerances written next to each assert (the north star asks for
parameter estimates within 1e-6 relative)."""
import numpy as np
import pytest

import helpers
from calico_amd import _capi, synthetic as syn

pytestmark = pytest.mark.gpu


def small_scene(camera_model=1, n_cameras=2, imu=True, imu_model=2, robust=False, noise=True, seed=7, **kw):
    return syn.make_scene(n_cameras, camera_model, imu, imu_model, cam_rate=10.0, imu_rate=50.0, duration=3.0,
                          segment_duration=3.0 / 23.9, pixel_noise=0.1 if noise else 0.0,
                          gyro_noise=1e-3 if noise else 0.0, accel_noise=1e-2 if noise else 0.0, robust=robust,
                          seed=seed, **kw)


def both(scene, hip, oracle):
    return syn.build_problem(hip, scene), syn.build_problem(oracle, scene)


def assert_eval_close(gpu, ref, rtol=1e-9):
    cg, gg, Hg = gpu.problem.evaluate()
    cr, gr, Hr = ref.problem.evaluate()
    assert Hg.shape == Hr.shape
    assert abs(cg - cr) <= rtol * abs(cr)
    # gradient / Gauss-Newton matrix: relative to the scale of the corresponding rows
    sg = np.sqrt(np.diag(Hr))
    sg = np.where(sg > 0, sg, 1.0)  # structurally zero columns (e.g. the gyroscope lever arm)
    assert np.abs(gg - gr).max() <= rtol * np.abs(gr).max()
    assert (np.abs(Hg - Hr) / np.outer(sg, sg)).max() <= rtol


@pytest.mark.parametrize("model", [1, 2, 3, 4, 5, 6, 7])
def test_camera_models_jtj_parity(model, hip, oracle):
    scene = small_scene(camera_model=model, imu=False, free_chart_pose=False)
    gpu, ref = both(scene, hip, oracle)
    assert_eval_close(gpu, ref)


def test_free_chart_pose_parity(hip, oracle):
    scene = small_scene(camera_model=1, imu=False, free_chart_pose=True)
    gpu, ref = both(scene, hip, oracle)
    assert_eval_close(gpu, ref)


def test_free_model_points_parity(hip, oracle):
    """model_definition_is_constant = false (world_model.cpp:52-61): every model point is a 3-vector block."""
    scene = small_scene(camera_model=1, n_cameras=2, imu=False, free_points=True)
    gpu, ref = both(scene, hip, oracle)
    assert_eval_close(gpu, ref)
    scene = small_scene(camera_model=3, n_cameras=1, imu=True, free_points=True, free_chart_pose=True)
    gpu, ref = both(scene, hip, oracle)
    assert_eval_close(gpu, ref)


def test_free_model_points_solve_matches_oracle(hip, oracle):
    scene = small_scene(camera_model=1, n_cameras=2, imu=True, free_points=True, seed=5)
    gpu, ref, sg, sr = solve_both(scene, hip, oracle, max_iter=25)
    assert sg.num_parameter_blocks_reduced == sr.num_parameter_blocks_reduced
    assert sg.num_effective_parameters_reduced == sr.num_effective_parameters_reduced
    ig, ir = gpu.problem.iterations(), ref.problem.iterations()
    assert len(ig) == len(ir)
    for a, b in zip(ig, ir):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-6 * abs(b.cost)
    assert sg.final_cost < sg.initial_cost * 1e-2
    assert_estimates_close(gpu, ref, scene)
    pg = np.stack([gpu.problem.get_param_block(int(b), 3) for b in gpu.point_blocks])
    pr = np.stack([ref.problem.get_param_block(int(b), 3) for b in ref.point_blocks])
    assert np.abs(pg - pr).max() <= 1e-6 * np.abs(pr).max()
    assert np.abs(pg - scene.points).max() > 0      # the free points did move


@pytest.mark.parametrize("imu_model", [1, 2, 3])
@pytest.mark.parametrize("robust", [False, True])
def test_imu_models_jtj_parity(imu_model, robust, hip, oracle):
    scene = small_scene(camera_model=1, n_cameras=1, imu=True, imu_model=imu_model, robust=robust)
    gpu, ref = both(scene, hip, oracle)
    assert_eval_close(gpu, ref)


def test_residual_writeback_and_inlier_mask(hip, oracle):
    scene = small_scene(camera_model=1, robust=True, outlier_fraction=0.05)
    gpu, ref = both(scene, hip, oracle)
    for i, s in enumerate(scene.sensors):
        rg, vg = gpu.problem.residuals(gpu.sensor_ids[i], s.n, s.dim)
        rr, vr = ref.problem.residuals(ref.sensor_ids[i], s.n, s.dim)
        assert np.array_equal(vg, vr)
        assert np.abs(rg - rr).max() <= 1e-9 * max(1.0, np.abs(rr).max())
        mg = gpu.problem.inlier_mask(gpu.sensor_ids[i], s.n, 3.0)
        mr = ref.problem.inlier_mask(ref.sensor_ids[i], s.n, 3.0)
        assert np.array_equal(mg, mr)  # integer mask: bit exact


def solve_both(scene, hip, oracle, max_iter=100, **opts):
    gpu, ref = both(scene, hip, oracle)
    og, orr = hip.default_options(), oracle.default_options()
    for o in (og, orr):
        o.minimizer_progress_to_stdout = 0
        o.max_num_iterations = max_iter
        o.num_threads = 8
        for k, v in opts.items():
            setattr(o, k, v)
    return gpu, ref, gpu.problem.solve(og), ref.problem.solve(orr)


def assert_estimates_close(gpu, ref, scene, rtol=1e-6):
    eg, cg = syn.read_back(gpu, scene)
    er, cr = syn.read_back(ref, scene)
    for a, b in zip(eg, er):
        for key in ("intrinsics", "t", "q"):
            scale = max(1e-3, np.abs(b[key]).max())
            assert np.abs(a[key] - b[key]).max() <= rtol * scale, key
        assert abs(a["latency"] - b["latency"]) <= rtol * max(1e-3, abs(b["latency"]))
    assert np.abs(cg - cr).max() <= rtol * max(1.0, np.abs(cr).max())


def test_toy_stereo_imu_solve_matches_oracle_and_truth(hip, oracle):
    """batch_optimizer_test.cpp:32-213 restated: perfect data, converge to truth within 1e-7."""
    # The toy problem is ill-conditioned (its own comment says so): most random draws of the
    # extrinsics converge within Ceres' default 50 iterations, a few wander off. Seed 4 is one of
    # the former (the reference's unseeded Eigen::Random draw evidently is too).
    scene = syn.make_scene(2, 1, True, 2, seed=4)
    gpu, ref, sg, sr = solve_both(scene, hip, oracle, max_iter=50)
    assert sg.termination_type == _capi.CONVERGENCE and sr.termination_type == _capi.CONVERGENCE
    assert sg.final_cost < 1e-7
    est, _ = syn.read_back(gpu, scene)
    for e, s in zip(est, scene.sensors):
        assert np.abs(e["intrinsics"] - s.intrinsics_true).max() < 1e-7
        assert np.abs(e["t"] - s.t_true).max() < 1e-7
        assert np.abs(e["q"] - s.q_true).max() < 1e-7
        assert abs(e["latency"] - s.latency_true) < 1e-7
    assert_estimates_close(gpu, ref, scene)
    assert sg.num_residual_blocks == sr.num_residual_blocks
    assert sg.num_residuals == sr.num_residuals
    assert sg.num_effective_parameters_reduced == sr.num_effective_parameters_reduced
    assert sg.num_parameters_reduced == sr.num_parameters_reduced


def test_noisy_robust_solve_matches_oracle(hip, oracle):
    scene = small_scene(camera_model=1, robust=True, outlier_fraction=0.02, seed=11)
    gpu, ref, sg, sr = solve_both(scene, hip, oracle, max_iter=60)
    assert sg.termination_type == sr.termination_type
    assert abs(sg.final_cost - sr.final_cost) <= 1e-8 * sr.final_cost
    assert_estimates_close(gpu, ref, scene)
    ig, ir = gpu.problem.iterations(), ref.problem.iterations()
    assert len(ig) == len(ir)
    for a, b in zip(ig, ir):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-6 * abs(b.cost)
    for i, s in enumerate(scene.sensors):
        if s.kind == _capi.SENSOR_CAMERA:
            mg = gpu.problem.inlier_mask(gpu.sensor_ids[i], s.n, 3.0)
            mr = ref.problem.inlier_mask(ref.sensor_ids[i], s.n, 3.0)
            assert np.array_equal(mg, mr)


@pytest.mark.parametrize("order", [4, 5, 7, 8])
def test_other_spline_orders(order, hip, oracle):
    """Orders other than the reference's default 6 take the generic kernels (no camera-frame path, run-time spline
    order in the evaluation, K-templated band kernels): same parity bar -- evaluation to 1e-9, solve to 1e-6."""
    scene = small_scene(camera_model=1, n_cameras=2, imu=True, imu_model=2, robust=True, order=order, seed=5)
    gpu, ref = both(scene, hip, oracle)
    assert_eval_close(gpu, ref)
    gpu, ref, sg, sr = solve_both(scene, hip, oracle, max_iter=30)
    assert sg.termination_type == sr.termination_type
    assert sg.num_iterations == sr.num_iterations
    assert abs(sg.final_cost - sr.final_cost) <= 1e-8 * sr.final_cost
    assert_estimates_close(gpu, ref, scene)


def test_project_matches_measurements_and_gives_zero_residuals(hip):
    """`calico_project` (Sensor::Project for the registered observations, camera.cpp:155-208, gyroscope.cpp:56-82,
    accelerometer.cpp:76-123) on the device: at the true parameters it reproduces the noise-free synthetic
    measurements (generated by the independent numpy restatement), and measurements generated WITH it give residuals
    that are exactly zero (PerfectDataPerfectResiduals, gyroscope_test.cpp:174-182 / accelerometer_test.cpp:194-202)."""
    scene = small_scene(camera_model=1, n_cameras=2, imu=True, imu_model=3, noise=False, perturb=False)
    built = syn.build_problem(hip, scene)
    P = built.problem
    preds = []
    for i, s in enumerate(scene.sensors):
        pred, valid = P.project(built.sensor_ids[i], s.n, s.dim)
        assert valid.all()
        scale = np.abs(s.meas).max()
        # The generator (like Sensor::Project) interpolates the pose at stamp - latency in ITS segment; the functor --
        # and therefore this prediction -- keeps the segment of the raw stamp and extrapolates (quirk Q3,
        # camera_cost_functor.cpp:13-14,52). They are the same number unless the latency straddles a knot.
        valid_knots = scene.knots[scene.order - 1: len(scene.knots) - scene.order + 1]
        seg = lambda t: np.clip(np.searchsorted(valid_knots, t, side="right") - 1, 0, len(valid_knots) - 2)
        same = seg(s.stamps) == seg(s.stamps - s.latency)
        assert same.sum() > 0.8 * s.n
        assert np.abs(pred - s.meas)[same].max() <= 1e-9 * scale      # numpy generator vs device kernels
        preds.append(pred)
    # feed the device's own predictions back as measurements: bit-exact zero residuals
    import copy
    scene2 = copy.deepcopy(scene)
    for s, pred in zip(scene2.sensors, preds):
        s.meas = pred.copy()
    built2 = syn.build_problem(hip, scene2)
    for i, s in enumerate(scene2.sensors):
        r, valid = built2.problem.residuals(built2.sensor_ids[i], s.n, s.dim)
        assert valid.all()
        assert np.all(r == 0.0)
    cost, _, _ = built2.problem.evaluate()
    # (the residual + Jacobian kernel contracts its multiply-adds differently from the residual-only one: rounding level)
    assert cost < 1e-18


# ---- the evaluation route for shapes outside the cell workgroups (camera.cpp:115-153 / accelerometer.cpp:35-56 add whatever
#      was measured: a 30 Hz camera on 10 Hz knots puts three frames into a spline segment, a 400 Hz IMU forty samples) ----
_ROUTE_SCENES = {
    # name: (cam_rate, imu_rate, what the plan must look like)
    "three_frames_per_cell": (30.0, 50.0, dict(max_frames_per_cell=3)),
    "several_items_per_imu_cell": (10.0, 400.0, dict(max_items_per_cell=2)),
    "both": (30.0, 400.0, dict(max_frames_per_cell=3, max_items_per_cell=2)),
}


def _route_scene(name, robust=True, seed=13, camera_model=1, imu_model=3, imu_rate=None):
    cam_rate, imu_rate_default, _ = _ROUTE_SCENES[name]
    imu_rate = imu_rate_default if imu_rate is None else imu_rate
    # knots at 10 Hz (make_scene's default) over 3 s: 0.1 s segments
    return syn.make_scene(2, camera_model, True, imu_model, cam_rate=cam_rate, imu_rate=imu_rate, duration=3.0,
                          segment_duration=3.0 / 23.9, pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=robust,
                          seed=seed)


def _assert_route(gpu, name):
    info = gpu.problem.plan_info()
    assert info["fuse_expand"] == 0, info             # eval_jacobian_kernel + expand_cells_kernel + row cells
    assert info["frames"] > 0 and info["items"] > 0, info
    for key, least in _ROUTE_SCENES[name][2].items():
        assert info[key] >= least, (key, info)
    return info


def test_default_shapes_take_the_cell_workgroups(hip):
    """The counterpart of the route tests below: the shapes every other parity test uses are evaluated by
    eval_cells_kernel (so together the two routes are both compared with the oracle)."""
    gpu = syn.build_problem(hip, small_scene(camera_model=1, n_cameras=2, imu=True, robust=True))
    info = gpu.problem.plan_info()
    assert info["fuse_expand"] == 1 and info["max_frames_per_cell"] <= 2 and info["tree_solver"] == 1, info


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("name", sorted(_ROUTE_SCENES))
def test_unfused_route_jtj_parity(name, robust, hip, oracle):
    """[cost, Jtr, JtJ] of the frame records + expansion launch + IMU row cells against the oracle, 1e-9."""
    scene = _route_scene(name, robust=robust)
    gpu, ref = both(scene, hip, oracle)
    _assert_route(gpu, name)
    assert_eval_close(gpu, ref)
    for i, s in enumerate(scene.sensors):
        rg, vg = gpu.problem.residuals(gpu.sensor_ids[i], s.n, s.dim)
        rr, vr = ref.problem.residuals(ref.sensor_ids[i], s.n, s.dim)
        assert np.array_equal(vg, vr)
        assert np.abs(rg - rr).max() <= 1e-9 * max(1.0, np.abs(rr).max())


@pytest.mark.parametrize("camera_model,imu_model", [(3, 2), (5, 1)])
def test_unfused_route_other_models(camera_model, imu_model, hip, oracle):
    scene = _route_scene("both", camera_model=camera_model, imu_model=imu_model, seed=17)
    gpu, ref = both(scene, hip, oracle)
    _assert_route(gpu, "both")
    assert_eval_close(gpu, ref)


@pytest.mark.parametrize("name", sorted(_ROUTE_SCENES))
def test_unfused_route_converged_solve_matches_oracle(name, hip, oracle):
    """Both sides to convergence with the default options: termination, iteration count, accept / reject sequence, cost per
    iteration, every estimate (1e-6) and the tau = 3 inlier masks (bit-exact)."""
    # (three_frames_per_cell: the IMU at 100 Hz here. At the shape's 50 Hz -- 150 samples per IMU for the twelve-parameter model --
    #  this draw leaves a nearly flat valley: with function / parameter tolerances of 1e-9 / 1e-10 the two sides end at the same
    #  cost with intrinsics 4e-4 apart, and with the default tolerances the estimates of three builds of this library sat at 0.6,
    #  0.8 and 1.1 of the 1e-6 bound, moved by last-bit changes of the linear solve -- such a minimum does not determine its
    #  estimates to 1e-6, whoever computes them. At 100 Hz (still one work item per IMU cell, three or four frames per camera
    #  cell) the same builds are at <= 0.12 of the bound: profiles/dev/route_diff2.py.)
    scene = _route_scene(name, seed=19, imu_rate=100.0 if name == "three_frames_per_cell" else None)
    gpu, ref, sg, sr = solve_both(scene, hip, oracle, max_iter=50)
    _assert_route(gpu, name)
    assert sg.termination_type == sr.termination_type == _capi.CONVERGENCE
    assert sg.num_iterations == sr.num_iterations
    ig, ir = gpu.problem.iterations(), ref.problem.iterations()
    assert [i.step_is_successful for i in ig] == [i.step_is_successful for i in ir]
    for a, b in zip(ig, ir):
        assert abs(a.cost - b.cost) <= 1e-6 * abs(b.cost)
    assert abs(sg.final_cost - sr.final_cost) <= 1e-8 * sr.final_cost
    assert_estimates_close(gpu, ref, scene)
    for i, s in enumerate(scene.sensors):
        if s.kind == _capi.SENSOR_CAMERA:
            assert np.array_equal(gpu.problem.inlier_mask(gpu.sensor_ids[i], s.n, 3.0),
                                  ref.problem.inlier_mask(ref.sensor_ids[i], s.n, 3.0))


def test_unfused_route_at_configs3_size(hip, oracle):
    """configs[3]'s rig with a 30 Hz camera and a 400 Hz IMU (the rates of a real rig; bench.py's shape is 20 / 200 Hz):
    ~150k blocks through the frame records, the expansion launch and four-item IMU row cells -- one evaluation to 1e-9 and
    three LM iterations against the oracle."""
    scene = syn.make_scene(4, 1, True, 3, cam_rate=30.0, imu_rate=400.0, duration=8.7, chart="april", seed=0xCA11C0 + 3,
                           pixel_noise=0.1, gyro_noise=1.7e-4 * np.sqrt(200.0), accel_noise=2e-3 * np.sqrt(200.0), robust=True,
                           segment_duration=8.7 / 23.9)
    gpu, ref = both(scene, hip, oracle)
    info = gpu.problem.plan_info()
    assert info["fuse_expand"] == 0 and info["max_frames_per_cell"] >= 3 and info["max_items_per_cell"] >= 2, info
    assert sum(s.n for s in scene.sensors) > 140000
    assert_eval_close(gpu, ref)
    gpu, ref, sg, sr = solve_both(scene, hip, oracle, max_iter=3, num_threads=64)
    ig, ir = gpu.problem.iterations(), ref.problem.iterations()
    assert len(ig) == len(ir)
    for a, b in zip(ig, ir):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-8 * abs(b.cost)
    assert_estimates_close(gpu, ref, scene)
