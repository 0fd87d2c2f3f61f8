"""N > 1 path on CPU: 2, 4 and 8 processes (gloo), each evaluating only its time window of the residual
blocks and all-reducing cost / gradient / JtJ, must walk the same LM iterations and reach the same
estimates as one process. The sharding rule is the product's (calico_amd/csrc/shard.hpp); the
collective plumbing (callback -> torch.distributed.all_reduce) is the one bench.py uses on RCCL."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene():
    from calico_amd import synthetic as syn
    return syn.make_scene(2, 1, True, 2, cam_rate=10.0, imu_rate=50.0, duration=3.0, segment_duration=3.0 / 23.9,
                          pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, max_cam_obs=1500, seed=3)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import helpers
    from calico_amd import synthetic as syn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = helpers.oracle_api()
    scene = _scene()
    built = syn.build_problem(api, scene)
    assert api.lib.oracle_problem_set_shard(built.problem.h, rank, world) == 0
    FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_int64)

    def allreduce(ctx, buf, n):
        arr = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(arr)
        dist.all_reduce(t)
        return 0
    cb = FN(allreduce)
    api.lib.oracle_problem_set_allreduce.argtypes = [C.c_void_p, FN, C.c_void_p]
    assert api.lib.oracle_problem_set_allreduce(built.problem.h, cb, None) == 0
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 25
    s = built.problem.solve(o)
    est, ctrl = syn.read_back(built, scene)
    its = [(i.iteration, i.step_is_successful, i.cost) for i in built.problem.iterations()]
    q.put((rank, s.final_cost, s.num_iterations, [e["intrinsics"] for e in est], ctrl, its))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_solve_equals_single_rank(world):
    """2, 4 and 8 ranks on the product's partition (contiguous time windows balanced by block count)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from calico_amd import synthetic as syn
    helpers.build_oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    api = helpers.oracle_api()
    scene = _scene()
    built = syn.build_problem(api, scene)
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 25
    s = built.problem.solve(o)
    est, ctrl = syn.read_back(built, scene)
    its = [(i.iteration, i.step_is_successful, i.cost) for i in built.problem.iterations()]
    for rank, cost, nit, intr, rctrl, rits in results:
        assert nit == s.num_iterations
        assert abs(cost - s.final_cost) <= 1e-9 * s.final_cost
        for a, b in zip(intr, [e["intrinsics"] for e in est]):
            assert np.allclose(a, b, rtol=1e-7, atol=1e-10)
        assert np.allclose(rctrl, ctrl, rtol=1e-7, atol=1e-9)
        assert [(a, b) for a, b, _ in rits] == [(a, b) for a, b, _ in its]
    # all ranks hold bit-identical estimates (same all-reduced system, same arithmetic)
    for r in results[1:]:
        assert np.array_equal(results[0][4], r[4])
