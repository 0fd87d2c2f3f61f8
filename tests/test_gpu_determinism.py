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


@pytest.mark.gpu
def test_solver_variants_agree(monkeypatch):
    """The tree solver (chains of any length at level 0), the sequential banded factorisation with and without its
    nested-dissection split, and the speculative evaluation are pure re-orderings of the same algorithm: whichever is
    selected (development toggles, read when the problem is finalised) the solve must take the same path -- same
    accept/reject sequence, estimates equal to rounding."""
    api = helpers.hip_api()
    scene = syn.make_scene(2, 1, True, 2, seed=4)      # 185 control points: 37 superblocks, several tree levels
    results = {}
    for solver, leaf, split, spec in [("bcr", "", "1", "1"), ("bcr", "1", "1", "1"), ("bcr", "3", "1", "0"), ("bcr", "8", "1", "1"),
                                      ("band", "", "1", "1"), ("band", "", "0", "1"), ("band", "", "1", "0"), ("band", "", "0", "0")]:
        monkeypatch.setenv("CALICO_SOLVER", solver)
        if leaf:
            monkeypatch.setenv("CALICO_BCR_LEAF", leaf)
        else:
            monkeypatch.delenv("CALICO_BCR_LEAF", raising=False)
        monkeypatch.setenv("CALICO_BAND_SPLIT", split)
        monkeypatch.setenv("CALICO_SPECULATIVE", spec)
        results[(solver, leaf, split, spec)] = _solve_repeatedly(api, scene, repeats=1, max_iter=50)[0]
    # the other development toggles (read per solve / per finalisation): blocking batches instead of the polled loop,
    # stand-alone control kernel, IMU items forming their own blocks, smaller IMU work items, the tree's top level
    # back-substituted in a launch of its own, the Schur complement / the first back-substitution in launches of their own,
    # the gather's source lists built on the host instead of by the device from the cell structure
    monkeypatch.delenv("CALICO_SOLVER")
    monkeypatch.delenv("CALICO_BCR_LEAF", raising=False)
    monkeypatch.setenv("CALICO_BAND_SPLIT", "1")
    monkeypatch.setenv("CALICO_SPECULATIVE", "1")
    for name, value in [("CALICO_STREAM_DEPTH", "0"), ("CALICO_FUSED_CONTROL", "0"), ("CALICO_ROW_CELLS", "0"),
                        ("CALICO_IMU_CHUNK", "7"), ("CALICO_STREAM_DEPTH", "1"), ("CALICO_BCR_MERGE_TOP", "0"),
                        ("CALICO_FUSE_SCHUR", "0"), ("CALICO_FUSE_BACK", "0"), ("CALICO_GATHER_STRUCT", "0"), ("CALICO_GATHER_TINY", "0"), ("CALICO_FOLD_FIRST", "0"),
                        ("CALICO_GATHER_FIXED", "0"), ("CALICO_ELIM", "panel"),       # round 4: the block factorisation of rounds 1-3 instead of block_elim.hpp
                        ("CALICO_PREDICT_END", "0"), ("CALICO_INLINE_NODES", "0"),
                        ("CALICO_FUSE_EXPAND", "0"),        # the cell expansion in a launch of its own, IMU cells as row cells
                        ("CALICO_GATHER_XCD", "0"), ("CALICO_SCHUR_SLICES", "2"), ("CALICO_HINT_FIRST", "0")]:        # the gather's workgroups in output order instead of dealt to the XCDs by stretches of time
        monkeypatch.setenv(name, value)
        results[(name, value)] = _solve_repeatedly(api, scene, repeats=1, max_iter=50)[0]
        monkeypatch.delenv(name)
    ref = results[("band", "", "0", "0")]
    assert ref[1] == _capi.CONVERGENCE
    for key, r in results.items():
        assert r[0] == ref[0] and r[1] == ref[1], key
        # noise-free toy problem: the cost goes to ~0, so late iterates only agree in absolute terms
        np.testing.assert_allclose(r[2], ref[2], rtol=1e-6, atol=1e-5, err_msg=str(key))
        np.testing.assert_allclose(r[3], ref[3], rtol=1e-6, atol=1e-9, err_msg=str(key))   # the north star's tolerance


@pytest.mark.gpu
def test_fused_dense_solve_and_back_substitution_launch_agrees(monkeypatch):
    """Round 3: on trees of up to two levels the first back-substitution launch rides in the launch of the dense reduced
    solve and takes its solution over an in-launch hand-off (sc1 stores + flag; dense_back_kernel). Same arithmetic in
    the same order: with the fusion switched off (two launches, CALICO_FUSE_BACK=0) the solve must walk the same
    iterations BIT FOR BIT -- a stale read behind the hand-off would show here (CALICO_BACK_PRE=0: the nodes as in the
    separate launch). Several chain lengths through CALICO_BCR_LEAF."""
    api = helpers.hip_api()
    # 4 cameras + IMU, robust kernels, ~90 control points like configs[3] (19 superblocks: two tree levels)
    scene = syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=8.7, chart="april", seed=21,
                           pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=8.7 / 23.9,
                           max_cam_obs=6000)
    for leaf in ("", "4", "9"):
        if leaf:
            monkeypatch.setenv("CALICO_BCR_LEAF", leaf)
        else:
            monkeypatch.delenv("CALICO_BCR_LEAF", raising=False)
        runs = {}
        for fuse, pre in (("1", "0"), ("0", "0"), ("1", "1")):
            monkeypatch.setenv("CALICO_FUSE_BACK", fuse)
            monkeypatch.setenv("CALICO_BACK_PRE", pre)
            runs[(fuse, pre)] = _solve_repeatedly(api, scene, repeats=3, max_iter=30)
        monkeypatch.delenv("CALICO_FUSE_BACK")
        monkeypatch.delenv("CALICO_BACK_PRE")
        ref = runs[("0", "0")][0]
        assert ref[0] > 3
        for key in (("1", "0"), ("0", "0")):
            for r in runs[key]:
                assert r[0] == ref[0] and r[1] == ref[1], (leaf, key)
                assert r[2] == ref[2], (leaf, key)
                assert np.array_equal(r[3], ref[3]), (leaf, key)
        # the nodes that form their solution as an affine map of the reduced solve's output BEFORE the hand-off
        # (back_node_pre, the default) associate the same products differently: equal to rounding, and reproducible
        pre = runs[("1", "1")]
        for r in pre:
            assert r[0] == ref[0] and r[1] == ref[1], leaf
            np.testing.assert_allclose(r[2], ref[2], rtol=1e-9)
            np.testing.assert_allclose(r[3], ref[3], rtol=1e-7, atol=1e-10)
            assert r[2] == pre[0][2] and np.array_equal(r[3], pre[0][3]), leaf


@pytest.mark.gpu
def test_back_to_back_solves_behind_leftover_kernels():
    """calico_solve returns as soon as the device reports the end of the solve; the early-exit kernels of the iterations
    enqueued ahead are still on the stream when the next call starts. Solves of different lengths, restarts, residual
    read-backs and evaluations issued back to back must each give what the same call gives on an idle stream (the
    progress words carry the solve's number, the results are written by the terminating stage itself)."""
    api = helpers.hip_api()
    scene = syn.make_scene(3, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=2.0, chart="april", seed=21,
                           pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=0.2)
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

    def run(max_iter, restart=True):
        if restart:
            P.set_param_blocks(ids, vals)
        o.max_num_iterations = max_iter
        s = P.solve(o)
        x = np.concatenate([np.asarray(P.get_param_block(int(b), n), float).ravel() for b, n in zip(ids, sizes)])
        return (s.num_iterations, s.termination_type, s.final_cost, tuple(float(i.cost) for i in P.iterations()), x)

    # references: every length once, with the stream drained before and after (calico_evaluate ends with a
    # synchronisation of the handle's stream)
    ref = {}
    for k in (0, 1, 2, 5, 30):
        P.evaluate()
        ref[k] = run(k)
        P.evaluate()
    # now back to back, in an order that puts short solves right behind long ones and vice versa
    order = [30, 0, 1, 30, 2, 1, 0, 5, 30, 5, 2, 2, 30, 1] * 3
    for k in order:
        got = run(k)
        assert got[0] == ref[k][0] and got[1] == ref[k][1]
        assert got[2] == ref[k][2] and got[3] == ref[k][3], "solve of %d iterations differs behind left-over kernels" % k
        assert np.array_equal(got[4], ref[k][4])
    # a continued solve (no restart) and calls of other kinds right behind a solve
    a = run(3)
    b = run(27, restart=False)
    assert b[2] <= a[2]
    r0, valid = P.residuals(built.sensor_ids[0], scene.sensors[0].n, scene.sensors[0].dim)
    cost, _, _ = P.evaluate()
    assert valid.all() and np.isfinite(r0).all()
    assert abs(cost - b[2]) <= 1e-9 * abs(b[2])
    # the epoch of the progress words wraps at 2047: cross it
    for _ in range(2100):
        got = run(1)
        assert got[0] == ref[1][0] and got[2] == ref[1][2]
    assert run(30)[3] == ref[30][3]


@pytest.mark.gpu
def test_large_reduced_system_kernels_agree(monkeypatch):
    """Rigs of many sensors have a reduced system of more than 128 columns (eight cameras + two IMUs: 220): it is factored
    panel by panel over several workgroups (reduced_block_step_mfma_kernel: the tree nodes' parts on the matrix cores),
    the remainder and the panels' backward sweep by the in-LDS 32-column-block solver. Round 2's kernels (64-row in-wave
    column Cholesky, VALU tile update, backward sweep in a launch of its own) stay behind CALICO_BLOCK_STEP=valu, the
    16-column panel solver for the remainder behind CALICO_DENSE=panel: the same algorithm in another order -- same
    iterations, estimates equal to rounding; and the default path is reproducible bit for bit."""
    api = helpers.hip_api()
    scene = syn.make_scene(8, 1, True, 3, cam_rate=5.0, imu_rate=50.0, duration=4.0, chart="april", seed=31,
                           pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=4.0 / 23.9,
                           max_cam_obs=20000, n_imus=2)
    runs = {}
    for step, dense, fused in (("", "", ""), ("valu", "", ""), ("", "panel", ""), ("", "", "1")):
        # (fused = "1": all steps and the in-LDS solver in ONE launch behind in-launch fan-ins, reduced_fused_kernel --
        #  measured 0.7 % slower than a launch per step at configs[4], hence off by default)
        if fused:
            monkeypatch.setenv("CALICO_REDUCED_FUSED", fused)
        else:
            monkeypatch.delenv("CALICO_REDUCED_FUSED", raising=False)
        if step:
            monkeypatch.setenv("CALICO_BLOCK_STEP", step)
        else:
            monkeypatch.delenv("CALICO_BLOCK_STEP", raising=False)
        if dense:
            monkeypatch.setenv("CALICO_DENSE", dense)
        else:
            monkeypatch.delenv("CALICO_DENSE", raising=False)
        runs[(step, dense, fused)] = _solve_repeatedly(api, scene, repeats=2, max_iter=20)
    monkeypatch.delenv("CALICO_REDUCED_FUSED", raising=False)
    ref = runs[("", "", "")]
    assert ref[0][0] > 3
    assert ref[0][2] == ref[1][2] and np.array_equal(ref[0][3], ref[1][3]), "the default path is not reproducible"
    for key, rr in runs.items():
        for r in rr:
            assert r[0] == ref[0][0] and r[1] == ref[0][1], key
            np.testing.assert_allclose(r[2], ref[0][2], rtol=1e-9, err_msg=str(key))
            np.testing.assert_allclose(r[3], ref[0][3], rtol=1e-7, atol=1e-9, err_msg=str(key))


@pytest.mark.gpu
def test_block_elimination_agrees_with_the_panel_factorisation(monkeypatch):
    """Round 4: every 32-column block of the tree levels and of the dense reduced solve is eliminated on the matrix cores
    (block_elim.hpp: a chief wave on the spine, follower waves with the identity rows and the rows of X, hand-over through a
    sentinel-filled LDS channel). Against the in-wave panel factorisation of rounds 1-3 (CALICO_ELIM=panel) -- a different
    order of the same sums -- the solve must take the same iterations with costs equal to 1e-9 and estimates to 1e-7, with
    and without the dense solve's launch fusion, for several chain lengths; and the default path must repeat bit for bit
    (a follower that read a channel entry too early would show as a run-to-run difference)."""
    api = helpers.hip_api()
    scene = syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=8.7, chart="april", seed=23,
                           pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=8.7 / 23.9,
                           max_cam_obs=6000)
    for leaf in ("", "2", "8"):
        if leaf:
            monkeypatch.setenv("CALICO_BCR_LEAF", leaf)
        else:
            monkeypatch.delenv("CALICO_BCR_LEAF", raising=False)
        runs = {}
        for elim, fuse in (("mfma", "1"), ("panel", "1"), ("mfma", "0")):
            monkeypatch.setenv("CALICO_ELIM", elim)
            monkeypatch.setenv("CALICO_FUSE_BACK", fuse)
            runs[(elim, fuse)] = _solve_repeatedly(api, scene, repeats=3 if elim == "mfma" else 1, max_iter=30)
        for reps in runs.values():
            for r in reps[1:]:
                assert r[0] == reps[0][0] and r[1] == reps[0][1]
                assert np.array_equal(r[2], reps[0][2]) and np.array_equal(r[3], reps[0][3])
        ref = runs[("panel", "1")][0]
        for key, reps in runs.items():
            r = reps[0]
            assert r[0] == ref[0] and r[1] == ref[1], (leaf, key)
            np.testing.assert_allclose(r[2], ref[2], rtol=1e-9, atol=0, err_msg=str((leaf, key)))
            np.testing.assert_allclose(r[3], ref[3], rtol=1e-7, atol=1e-10, err_msg=str((leaf, key)))
    monkeypatch.delenv("CALICO_ELIM")
    monkeypatch.delenv("CALICO_FUSE_BACK")


@pytest.mark.gpu
def test_end_prediction_and_inline_descriptors_change_nothing(monkeypatch):
    """Round 4 (second half): (a) the Jacobian launch of every iteration tells the host whether the control stage behind
    it is about to end the solve, and the host enqueues the next iteration on that word instead of always one ahead
    (CALICO_PREDICT_END); (b) the tree levels get their node descriptors without a load -- level 0's by arithmetic, the
    top levels' with the kernel arguments --, request their first block from both reduce buffers before the LM state has
    arrived, and read the state with vector loads (CALICO_INLINE_NODES=0 restores the table look-up). Neither touches
    the arithmetic: solves of every length -- converged, cut by the budget at every iteration count, restarted back to
    back -- must be bit-identical with the switches off. A wrong hint shows as a hang or a lost iteration, a wrong
    descriptor or buffer as different numbers."""
    api = helpers.hip_api()
    scene = syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=8.7, chart="april", seed=21,
                           pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=8.7 / 23.9,
                           max_cam_obs=6000)
    long_scene = syn.make_scene(2, 1, True, 2, seed=4)       # 185 control points: several tree levels, the middle ones read the table
    for sc, budgets in ((scene, (50, 1, 2, 3, 5, 8)), (long_scene, (50, 4))):
        runs = {}
        for pe, inl in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
            monkeypatch.setenv("CALICO_PREDICT_END", pe)
            monkeypatch.setenv("CALICO_INLINE_NODES", inl)
            runs[(pe, inl)] = [_solve_repeatedly(api, sc, repeats=2, max_iter=b) for b in budgets]
        monkeypatch.delenv("CALICO_PREDICT_END")
        monkeypatch.delenv("CALICO_INLINE_NODES")
        ref = runs[("0", "0")]
        assert ref[0][0][1] == _capi.CONVERGENCE and ref[0][0][0] > 3
        for key, per_budget in runs.items():
            for rb, r in zip(ref, per_budget):
                for a, b in zip(rb, r):
                    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], key
                    assert np.array_equal(a[3], b[3]), key


@pytest.mark.gpu
def test_cell_workgroups_equal_the_expansion_launch(monkeypatch):
    """Round 4: where every camera cell has at most two frames and every IMU cell is one work item, the Jacobian launch
    runs workgroups of two waves (eval_cells_kernel): the frames of a cell leave M_ext and the expansion coefficients in
    LDS and expand the cell's block together behind one workgroup barrier -- no compact record, no expand_cells_kernel
    launch --, and IMU items form their blocks themselves. Every block is the sum the separate expansion launch forms, in
    the same order (configs[1..3], where every cell has two frames, come out bit for bit: profiles/dev/bitwise.py); one-frame
    cells are packed two to a workgroup (each wave expands its own), which reorders the frames' cost slots and so regroups
    the sum of the costs -- so against CALICO_FUSE_EXPAND=0 (records + expansion launch + row cells) the iterations must
    agree to rounding, and each variant must reproduce itself bit for bit. Scenes: two frames per cell (and a few with
    one at the ends); one frame per cell (10 Hz camera); three frames in some cells (the plan falls back to the launch of
    its own)."""
    api = helpers.hip_api()
    common = dict(chart="april", pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, max_cam_obs=6000)
    scenes = [
        syn.make_scene(4, 1, True, 3, cam_rate=20.0, imu_rate=100.0, duration=4.0, seed=31, segment_duration=4.0 / 23.9, **common),
        syn.make_scene(2, 3, True, 2, cam_rate=10.0, imu_rate=100.0, duration=4.0, seed=32, segment_duration=4.0 / 23.9, **common),
        syn.make_scene(2, 1, True, 3, cam_rate=30.0, imu_rate=100.0, duration=3.0, seed=33, segment_duration=3.0 / 23.9, **common),
    ]
    for sc in scenes:
        runs = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("CALICO_FUSE_EXPAND", fuse)
            runs[fuse] = _solve_repeatedly(api, sc, repeats=3, max_iter=20)
        monkeypatch.delenv("CALICO_FUSE_EXPAND")
        ref = runs["0"][0]
        assert ref[0] > 3
        for fuse in ("1", "0"):
            first = runs[fuse][0]
            for r in runs[fuse][1:]:
                assert r[0] == first[0] and r[1] == first[1] and r[2] == first[2] and np.array_equal(r[3], first[3]), fuse
            assert first[0] == ref[0] and first[1] == ref[1]
            np.testing.assert_allclose(first[2], ref[2], rtol=1e-9)
            np.testing.assert_allclose(first[3], ref[3], rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_chain_look_ahead_is_bit_identical(monkeypatch):
    """Round 5: the tree levels' chain steps with look-ahead (bcr_level_kernel<.., LA>, CALICO_LOOKAHEAD=1: the chief starts on
    the next block of a chain behind the diagonal's Schur update alone, the followers form the update of their own input
    tiles in registers, Z is double-buffered, the loader waves carry the left separator's sums) reorder WHO computes a
    product and WHEN, never the products or the order of a sum: every iteration, cost and estimate must equal the default
    path's bit for bit -- chains of 4 (configs[3]'s shape), of 8 (long trajectory, several levels) and of 2 / 3 (leaf
    override), border roles and role 0 alike; three repeats each (a hazard between a step's LDS buffers would show as a
    run-to-run difference)."""
    api = helpers.hip_api()
    common = dict(chart="april", pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, max_cam_obs=6000)
    scenes = [
        (syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=8.7, seed=41, segment_duration=8.7 / 23.9, **common), None),
        (syn.make_scene(2, 1, True, 2, seed=4), None),        # 185 control points: several tree levels
        (syn.make_scene(2, 3, True, 2, cam_rate=10.0, imu_rate=100.0, duration=6.0, seed=42, segment_duration=6.0 / 23.9, **common), "2"),
        (syn.make_scene(2, 3, True, 2, cam_rate=10.0, imu_rate=100.0, duration=6.0, seed=42, segment_duration=6.0 / 23.9, **common), "3"),
    ]
    for sc, leaf in scenes:
        if leaf:
            monkeypatch.setenv("CALICO_BCR_LEAF", leaf)
        runs = {}
        for la in ("0", "1"):
            monkeypatch.setenv("CALICO_LOOKAHEAD", la)
            runs[la] = _solve_repeatedly(api, sc, repeats=3, max_iter=20)
        monkeypatch.delenv("CALICO_LOOKAHEAD")
        if leaf:
            monkeypatch.delenv("CALICO_BCR_LEAF")
        ref = runs["0"][0]
        assert ref[0] > 3
        for la in ("0", "1"):
            for r in runs[la]:
                assert r[0] == ref[0] and r[1] == ref[1] and r[2] == ref[2], (la, leaf)
                assert np.array_equal(r[3], ref[3]), (la, leaf)


@pytest.mark.gpu
def test_rolling_chief_equals_the_barrier_form_bit_for_bit(monkeypatch):
    """Round 6: level 0 of the tree solver eliminates its chains with a ROLLING CHIEF (bcr_level_kernel<.., ROLL>): waves 0 and 1
    take turns as the chief, the follower forms the next block's diagonal D_{k+1} - Z^BᵀZ^B step by step in the chief's own
    register layout and goes on as its chief, tiles come from R(x) through a per-lane offset table, later blocks' inputs are
    staged as tile images a block ahead, and nothing waits at a workgroup barrier -- order is kept by single-writer counters
    and write-once channels in LDS. The products and their order are those of the barrier form (CALICO_ROLL=0), so every
    iterate must come out BIT FOR BIT the same -- for chains of 1, 2, 3, 4 and 8 blocks (odd / even hand-overs, the channel
    ring of three wrapping, the image ring of two wrapping), a trajectory whose last superblock is partly padding, spline orders
    4 and 5 (other band structure in the offset table), a trajectory with unobserved control points, with and without the same form on the upper levels
    (CALICO_ROLL_UPPER=1) -- and repeat itself run to run (a follower that read a tile image, a channel entry or a Z row too
    early, or a buffer reused too soon, shows as a run-to-run difference or a difference to the barrier form)."""
    api = helpers.hip_api()
    common = dict(chart="april", pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, max_cam_obs=6000)
    scenes = [
        ("92 control points", syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=8.7, seed=41, segment_duration=8.7 / 23.9, **common), ("", "1", "2", "3", "8")),
        ("185 control points", syn.make_scene(2, 1, True, 2, seed=4), ("", "3")),
        ("ragged end", syn.make_scene(2, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=7.3, seed=43, segment_duration=7.3 / 23.9, **common), ("", "2")),
    ]
    # data over the first 5 s of a trajectory of 8.7 s: the control points behind it are unobserved -- padding rows in the tiles (a
    # five-bit activity mask per superblock)
    unobserved = syn.make_scene(2, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=5.0, seed=62, segment_duration=8.7 / 23.9, **common)
    assert syn.build_problem(api, unobserved).problem.plan_info()["all_control_points_observed"] == 0
    scenes.append(("unobserved control points", unobserved, ("", "2", "3")))
    for order in (4, 5):
        scenes.append(("order %d" % order, syn.make_scene(2, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=6.0, seed=44 + order, segment_duration=6.0 / 23.9,
                                                           order=order, **common), ("",)))
    for name, sc, leaves in scenes:
        for leaf in leaves:
            if leaf:
                monkeypatch.setenv("CALICO_BCR_LEAF", leaf)
            else:
                monkeypatch.delenv("CALICO_BCR_LEAF", raising=False)
            runs = {}
            for roll, upper in (("0", "0"), ("1", "0"), ("1", "1")):
                monkeypatch.setenv("CALICO_ROLL", roll)
                monkeypatch.setenv("CALICO_ROLL_UPPER", upper)
                runs[(roll, upper)] = _solve_repeatedly(api, sc, repeats=3 if roll == "1" else 1, max_iter=12)
            monkeypatch.delenv("CALICO_ROLL")
            monkeypatch.delenv("CALICO_ROLL_UPPER")
            ref = runs[("0", "0")][0]
            assert ref[0] >= 3, (name, leaf)
            for key, reps in runs.items():
                for r in reps:
                    assert r[0] == ref[0] and r[1] == ref[1] and r[2] == ref[2], (name, leaf, key)
                    assert np.array_equal(r[3], ref[3]), (name, leaf, key)
    monkeypatch.delenv("CALICO_BCR_LEAF", raising=False)


@pytest.mark.gpu
def test_rolling_owners_of_the_dense_solve_agree_with_the_barrier_form(monkeypatch):
    """Round 6: the dense reduced solve (<= 128 unknowns) eliminates its 32-column blocks with ROLLING OWNERS (dense_block_solve_body,
    elim == 2): wave j owns block-row j for the whole solve, keeps its diagonal block D_j in registers (the chief's layout) and
    subtracts Z(j,l) Z(j,l)ᵀ step by step while it follows every block l < j, accumulates what block l does to the rows it follows
    block l+1 with beside it (the other operand read from the next chief's rows behind a progress word), and goes on as block j's
    chief when block j-1's last pivot is through -- no barrier between the blocks, two compact channels, single-writer counters.
    The sums are those of the barrier form (CALICO_DENSE_ROLL=0) in another order: same iterations, costs to 1e-9, estimates to
    1e-7 -- for reduced systems of one, two, three and four blocks (1 camera: 39 unknowns; stereo + IMU; 4 cameras + IMU: 122),
    with the dense solve in a launch of its own (CALICO_FUSE_BACK=0) and riding with the back-substitution -- and the default
    must repeat itself bit for bit (a wave that read a row, a channel entry or a progress word too early shows run to run)."""
    api = helpers.hip_api()
    common = dict(chart="april", pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, max_cam_obs=6000)
    scenes = [
        syn.make_scene(1, 1, False, cam_rate=10.0, duration=6.0, seed=51, segment_duration=6.0 / 23.9, chart="april", pixel_noise=0.1, max_cam_obs=6000),
        syn.make_scene(2, 3, True, 2, cam_rate=10.0, imu_rate=100.0, duration=6.0, seed=52, segment_duration=6.0 / 23.9, **common),
        syn.make_scene(3, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=6.0, seed=53, segment_duration=6.0 / 23.9, **common),
        syn.make_scene(4, 1, True, 3, cam_rate=10.0, imu_rate=100.0, duration=8.7, seed=54, segment_duration=8.7 / 23.9, **common),
    ]
    for i, sc in enumerate(scenes):
        runs = {}
        for roll, fuse in (("0", "1"), ("1", "1"), ("1", "0")):
            monkeypatch.setenv("CALICO_DENSE_ROLL", roll)
            monkeypatch.setenv("CALICO_FUSE_BACK", fuse)
            runs[(roll, fuse)] = _solve_repeatedly(api, sc, repeats=3 if roll == "1" else 1, max_iter=15)
        monkeypatch.delenv("CALICO_DENSE_ROLL")
        monkeypatch.delenv("CALICO_FUSE_BACK")
        ref = runs[("0", "1")][0]
        assert ref[0] >= 3, i
        for key, reps in runs.items():
            first = reps[0]
            for r in reps[1:]:
                assert r[0] == first[0] and r[1] == first[1] and r[2] == first[2] and np.array_equal(r[3], first[3]), (i, key)
            assert first[0] == ref[0] and first[1] == ref[1], (i, key)
            np.testing.assert_allclose(first[2], ref[2], rtol=1e-9, atol=0, err_msg=str((i, key)))
            # (absolute part: the weakly determined small components -- rotation-vector entries of 1e-4 -- move by a few 1e-10
            #  between two orders of the same sums after fifteen iterations)
            np.testing.assert_allclose(first[3], ref[3], rtol=1e-7, atol=2e-9, err_msg=str((i, key)))
