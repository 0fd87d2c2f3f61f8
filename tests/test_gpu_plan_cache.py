"""The plan cache of libcalico_hip.so (include/calico_hip.h, calico_plan_cache_*): the reference rebuilds its problem on
every Optimize() (batch_optimizer.cpp:57-70); a rebuilt problem whose STRUCTURE the library has seen before adopts the
cached plan and a pooled workspace and only uploads its values -- and must behave exactly like a freshly planned one.
Any change of structure must miss the cache."""
import copy

import numpy as np
import pytest

from calico_amd import _capi, synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(seed=11):
    return syn.make_scene(2, 1, True, 2, cam_rate=10.0, imu_rate=50.0, duration=3.0, segment_duration=3.0 / 23.9,
                          pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, seed=seed)


def _solve(built, api, n=25):
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = n
    s = built.problem.solve(o)
    return s, [(i.iteration, i.step_is_successful, i.cost) for i in built.problem.iterations()]


def test_identical_structure_hits_and_solves_bit_identically(hip):
    hip.plan_cache_clear()
    scene = _scene()
    h0, m0, _ = _capi.plan_cache_stats(hip)
    a = syn.build_problem(hip, scene)
    sa, ita = _solve(a, hip)
    ea, ca = syn.read_back(a, scene)
    h1, m1, n1 = _capi.plan_cache_stats(hip)
    assert (h1 - h0, m1 - m0) == (0, 1) and n1 == 1
    # a second handle while the first is alive (shared structure buffers, its own workspace) ...
    b = syn.build_problem(hip, scene)
    sb, itb = _solve(b, hip)
    eb, cb = syn.read_back(b, scene)
    h2, m2, _ = _capi.plan_cache_stats(hip)
    assert (h2 - h1, m2 - m1) == (1, 0)
    assert itb == ita and np.array_equal(ca, cb)
    # ... and a third after both are gone: plan AND workspace recycled (the reference's Optimize-per-call pattern)
    a.problem.close()
    b.problem.close()
    c = syn.build_problem(hip, scene)
    sc, itc = _solve(c, hip)
    ec, cc = syn.read_back(c, scene)
    assert _capi.plan_cache_stats(hip)[0] - h2 == 1
    assert itc == ita and np.array_equal(ca, cc)
    for x, y in zip(ea, ec):
        assert np.array_equal(x["intrinsics"], y["intrinsics"]) and np.array_equal(x["q"], y["q"])
    # the residual write-back and the tagging path work on the recycled workspace, too
    for sid, sp in zip(c.sensor_ids, scene.sensors):
        r, valid = c.problem.residuals(sid, sp.n, sp.dim)
        assert valid.all() and np.isfinite(r).all()


def test_new_values_same_structure_hit_and_match_a_fresh_plan(hip, oracle):
    """Other measurements and another starting point on a known structure: served from the cache, equal to the oracle."""
    hip.plan_cache_clear()
    scene = _scene()
    first = syn.build_problem(hip, scene)
    _solve(first, hip, 3)
    first.problem.close()
    other = copy.deepcopy(scene)
    rng = np.random.default_rng(5)
    for s in other.sensors:
        s.meas = s.meas + 0.05 * rng.standard_normal(s.meas.shape)
        s.intrinsics = s.intrinsics * (1.0 + 1e-3 * rng.standard_normal(s.intrinsics.shape))
    other.ctrl = other.ctrl + 1e-4 * rng.standard_normal(other.ctrl.shape)
    h0, m0, _ = _capi.plan_cache_stats(hip)
    g = syn.build_problem(hip, other)
    cg, gg, Hg = g.problem.evaluate()
    assert _capi.plan_cache_stats(hip)[:2] == (h0 + 1, m0)
    r = syn.build_problem(oracle, other)
    cr, gr, Hr = r.problem.evaluate()
    assert abs(cg - cr) <= 1e-10 * abs(cr)
    assert np.abs(gg - gr).max() <= 1e-9 * np.abs(gr).max()
    sd = np.sqrt(np.diag(Hr)); sd = np.where(sd > 0, sd, 1.0)
    assert (np.abs(Hg - Hr) / np.outer(sd, sd)).max() <= 1e-9
    sg, itg = _solve(g, hip, 10)
    sr, itr = _solve(r, oracle, 10)
    assert [(a, b) for a, b, _ in itg] == [(a, b) for a, b, _ in itr]
    for (_, _, x), (_, _, y) in zip(itg, itr):
        assert abs(x - y) <= 1e-6 * abs(y)


@pytest.mark.parametrize("change", ["stamp", "constant", "loss", "sigma", "drop_observation", "point"])
def test_changed_structure_misses_the_cache(change, hip, oracle):
    hip.plan_cache_clear()
    scene = _scene()
    base = syn.build_problem(hip, scene)
    base.problem.finalize()
    h0, m0, _ = _capi.plan_cache_stats(hip)
    other = copy.deepcopy(scene)
    cam = other.sensors[0]
    if change == "stamp":              # one frame a hair later: other frames / segments for its observations
        t0 = cam.stamps[0]
        cam.stamps = np.where(cam.stamps == t0, t0 + 1e-3, cam.stamps)
    elif change == "constant":
        cam.enable_intrinsics = not cam.enable_intrinsics
    elif change == "loss":
        cam.loss = 1 if cam.loss != 1 else 2
    elif change == "sigma":
        cam.sigma = cam.sigma * 2.0
    elif change == "drop_observation":
        keep = np.ones(cam.n, bool); keep[7] = False
        cam.meas, cam.stamps, cam.point_idx = cam.meas[keep], cam.stamps[keep], cam.point_idx[keep]
    elif change == "point":
        cam.point_idx = cam.point_idx.copy(); cam.point_idx[3] = (cam.point_idx[3] + 1) % len(other.points)
    g = syn.build_problem(hip, other)
    cg, gg, Hg = g.problem.evaluate()
    assert _capi.plan_cache_stats(hip)[:2] == (h0, m0 + 1)          # planned afresh, not served from the cache
    r = syn.build_problem(oracle, other)
    cr, gr, Hr = r.problem.evaluate()
    assert Hg.shape == Hr.shape
    assert abs(cg - cr) <= 1e-10 * abs(cr)
    assert np.abs(gg - gr).max() <= 1e-9 * np.abs(gr).max()
    # the unchanged structure is still served
    again = syn.build_problem(hip, scene)
    again.problem.finalize()
    assert _capi.plan_cache_stats(hip)[0] == h0 + 1


def test_setup_cost_of_a_known_structure(hip):
    """What the cache is for: on a structure seen before, build + finalize of a fresh handle is a matter of value uploads."""
    import time
    hip.plan_cache_clear()
    scene = syn.config_scene(3)
    t = []
    for _ in range(4):
        t0 = time.perf_counter()
        b = syn.build_problem(hip, scene)
        t1 = time.perf_counter()
        b.problem.finalize()
        t2 = time.perf_counter()
        t.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        b.problem.close()
    print("configs[3] add_* / finalize ms: first %s, then %s" % (t[0], t[1:]))
    assert min(x[1] for x in t[1:]) < 0.5 * t[0][1]


def test_device_memory_goes_back_to_the_driver(hip):
    """The library's allocation slabs (64 MB each; a buffer larger than a slab is an allocation of its own) are not kept for
    good: destroying the handles leaves at most one idle slab per device, calico_plan_cache_clear() none -- a long-lived
    process that shares the GPU with PyTorch / RCCL, or that once solved a large problem, gets its memory back."""
    import torch
    MB = 1 << 20

    def cycle(scene):
        b = syn.build_problem(hip, scene)
        _solve(b, hip, n=3)
        b.problem.close()

    small = _scene(seed=23)
    # configs[3]'s rig over 3 s: ~36k blocks, partials + workspaces beyond one slab
    large = syn.make_scene(4, 1, True, 3, cam_rate=20.0, imu_rate=200.0, duration=3.0, chart="april", seed=29, pixel_noise=0.1,
                           gyro_noise=1e-3, accel_noise=1e-2, robust=True, segment_duration=3.0 / 23.9)
    cycle(small)                      # code objects, streams, pinned pools: what a process keeps whatever it does
    cycle(large)
    hip.plan_cache_clear()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    held = syn.build_problem(hip, large)
    _solve(held, hip, n=3)
    in_use = free0 - torch.cuda.mem_get_info(0)[0]
    assert in_use >= 16 * MB          # (the test would prove nothing otherwise)
    held.problem.close()
    del held
    cycle(small)
    hip.plan_cache_clear()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 <= 4 * MB, (free0, free1, in_use)
    # ... and with the cache left alone, handles come and go without the footprint growing
    cycle(large)
    base = torch.cuda.mem_get_info(0)[0]
    for _ in range(3):
        cycle(large)
        cycle(small)
    assert base - torch.cuda.mem_get_info(0)[0] <= 4 * MB
    hip.plan_cache_clear()
