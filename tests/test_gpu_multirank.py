"""The N > 1 path of libcalico_hip.so on one GPU box.

1. world = 1 with the all-reduce callback installed: the multi-rank control flow (fixed-address collective target,
   commit by copy, iterations enqueued in batches) must walk exactly the iterations of the plain single-rank solve.
2. Two processes sharing the one GPU, each evaluating its own time window of residual blocks (shard.hpp), exchanging
   [cost | Jtr | JtJ] through the callback (device buffer -> host -> gloo all-reduce -> device buffer): both ranks must
   reach the estimates of the single-rank solve (1e-9 relative: the sum is associated differently), in the batched
   mode and in the one-iteration-per-round-trip mode (CALICO_MULTIRANK_ASYNC=0).
3. The native exchange (calico_comm_init_rccl: the handle's own RCCL communicator, what bench.py uses on several GPUs)
   with a world of one rank; and the refusal of a sharded handle that has no exchange at all."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _scene():
    from calico_amd import synthetic as syn
    return syn.make_scene(2, 1, True, 2, cam_rate=10.0, imu_rate=50.0, duration=3.0, segment_duration=3.0 / 23.9,
                          pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, seed=3)


def _solve(P, api, sync_every):
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 25
    o.sync_every = sync_every
    s = P.solve(o)
    return s, [(i.iteration, i.step_is_successful, i.cost) for i in P.iterations()]


def test_native_rccl_exchange_equals_plain_solve(hip):
    """The production exchange: the handle owns an RCCL communicator (calico_comm_init_rccl) and all-reduces natively on
    its own stream. With a world of one rank the all-reduce is the identity, so the solve must walk exactly the
    iterations of the plain one, bit for bit -- through the real ncclAllReduce calls, no Python in the loop."""
    from calico_amd import _capi, synthetic as syn
    scene = _scene()
    plain = syn.build_problem(hip, scene)
    s0, it0 = _solve(plain.problem, hip, 8)
    coll = syn.build_problem(hip, scene)
    coll.problem.comm_init_rccl(_capi.comm_unique_id(hip), 0, 1)
    s1, it1 = _solve(coll.problem, hip, 8)
    assert s1.termination_type == s0.termination_type and s1.num_iterations == s0.num_iterations
    assert it0 == it1            # same kernels on the same data: bit-identical costs
    e0, _ = syn.read_back(plain, scene)
    e1, _ = syn.read_back(coll, scene)
    for a, b in zip(e0, e1):
        assert np.array_equal(a["intrinsics"], b["intrinsics"])


def test_shard_without_exchange_is_refused(hip):
    """A rank of a larger world with no way to exchange would silently solve its own time window only
    (ADVICE r1): solve and evaluate refuse instead."""
    from calico_amd import _capi, synthetic as syn
    built = syn.build_problem(hip, _scene())
    built.problem.set_shard(0, 2)
    o = hip.default_options()
    o.minimizer_progress_to_stdout = 0
    with pytest.raises(_capi.CalicoError) as e:
        built.problem.solve(o)
    assert e.value.code == _capi.FAILED_PRECONDITION
    with pytest.raises(_capi.CalicoError) as e:
        built.problem.evaluate()
    assert e.value.code == _capi.FAILED_PRECONDITION
    built.problem.set_shard(0, 1)       # back to a world of one: fine again
    assert built.problem.solve(o).num_iterations > 0


def test_collective_control_flow_equals_plain_solve(hip):
    from calico_amd import synthetic as syn
    scene = _scene()
    plain = syn.build_problem(hip, scene)
    s0, it0 = _solve(plain.problem, hip, 8)
    calls = []

    def allreduce(ctx, buf, n, strm):   # one rank: the sum over ranks is the buffer itself
        calls.append(n)
        return 0
    coll = syn.build_problem(hip, scene)
    coll.problem.set_shard(0, 1)
    coll.problem.set_allreduce(allreduce)
    s1, it1 = _solve(coll.problem, hip, 8)
    assert len(calls) >= s1.num_iterations
    assert s1.termination_type == s0.termination_type and s1.num_iterations == s0.num_iterations
    assert [(a, b) for a, b, _ in it0] == [(a, b) for a, b, _ in it1]
    for (_, _, c0), (_, _, c1) in zip(it0, it1):
        assert c0 == c1          # same kernels on the same data: bit-identical costs
    e0, c0 = syn.read_back(plain, scene)
    e1, c1 = syn.read_back(coll, scene)
    assert np.array_equal(c0, c1)
    for a, b in zip(e0, e1):
        assert np.array_equal(a["intrinsics"], b["intrinsics"])


def _worker(rank, world, port, q, batched):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["CALICO_MULTIRANK_ASYNC"] = "1" if batched else "0"
    import torch
    import torch.distributed as dist
    import helpers
    from calico_amd import synthetic as syn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")   # torch's HIP runtime first, as in bench.py; the library then shares it
    api = helpers.hip_api()
    scene = _scene()
    built = syn.build_problem(api, scene)
    P = built.problem
    P.set_stream(torch.cuda.current_stream().cuda_stream)
    P.set_shard(rank, world)

    def allreduce(ctx, buf, n, strm):
        t = torch.as_tensor(_DevArray(buf, n), device="cuda")
        h = t.cpu()              # waits for the kernels enqueued so far on the problem's stream
        dist.all_reduce(h)
        t.copy_(h)
        return 0
    P.set_allreduce(allreduce)
    s, its = _solve(P, api, 4)
    est, ctrl = syn.read_back(built, scene)
    q.put((rank, s.final_cost, s.num_iterations, s.termination_type, [e["intrinsics"] for e in est], ctrl, its))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batched", [True, False])
def test_two_ranks_on_one_gpu_equal_single_rank(batched, hip):
    import multiprocessing as mp
    from calico_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (1 if batched else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, batched)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene = _scene()
    single = syn.build_problem(hip, scene)
    s, its = _solve(single.problem, hip, 4)
    est, ctrl = syn.read_back(single, scene)
    for rank, final_cost, n_it, term, intr, c, rits in results:
        assert term == s.termination_type and n_it == s.num_iterations
        assert abs(final_cost - s.final_cost) <= 1e-9 * s.final_cost
        assert [(a, b) for a, b, _ in rits] == [(a, b) for a, b, _ in its]
        for (_, _, c0), (_, _, c1) in zip(its, rits):
            assert abs(c0 - c1) <= 1e-9 * abs(c0)
        assert np.abs(c - ctrl).max() <= 1e-9 * np.abs(ctrl).max()
        for a, b in zip(intr, est):
            assert np.abs(a - b["intrinsics"]).max() <= 1e-9 * np.abs(b["intrinsics"]).max()
    # both ranks hold the same estimates bit for bit (deterministic all-reduce, replicated solve)
    assert np.array_equal(results[0][5], results[1][5])


def test_two_handles_one_device_stream_ordered_exchange_at_configs3_size(hip, monkeypatch):
    """The buffer choreography of the native multi-rank path -- zero-filled exchange target, the rank's gather into it, the
    exchange on the handle's stream, commit_kernel for an accepted candidate, iterations enqueued ahead of the device in
    batches -- with the semantics of a world of TWO at the size of BASELINE configs[3], on one GPU: rank 0 and rank 1 are two
    handles of this process on streams of their own, each driven by its own thread, and the exchange is a stream-ordered
    stand-in for ncclAllReduce (nothing in it waits on the host for the device: copy to a staging buffer, event, wait for
    the peer's event, sum in rank order -- so that, unlike the two-process test above whose callback blocks on a device-to-
    host copy, the library's asynchronous loop really runs ahead of the kernels). Both ranks must walk the single-rank
    solve's iterations and agree with each other bit for bit. (RCCL itself refuses two ranks on one device.)"""
    import threading
    import torch
    from calico_amd import synthetic as syn
    monkeypatch.setenv("CALICO_MULTIRANK_ASYNC", "1")
    scene = syn.config_scene(3)
    world = 2
    streams = [torch.cuda.Stream() for _ in range(world)]
    staged = [dict() for _ in range(world)]       # rank -> call number -> (staging tensor, event): kept alive to the end
    calls = [0] * world
    meet = threading.Barrier(world, timeout=120)
    lock = threading.Lock()
    errors = []

    def make_allreduce(rank):
        def allreduce(ctx, buf, n, strm):
            try:
                k = calls[rank]
                calls[rank] += 1
                assert strm == streams[rank].cuda_stream
                with torch.cuda.stream(streams[rank]):
                    mine = torch.as_tensor(_DevArray(buf, n), device="cuda")
                    stage = mine.clone()
                    ev = torch.cuda.Event()
                    ev.record(streams[rank])
                    with lock:
                        staged[rank][k] = (stage, ev)
                    meet.wait()                      # both ranks have ENQUEUED call k (a rank that issued fewer calls: timeout)
                    with lock:
                        parts = [staged[r][k] for r in range(world)]
                    streams[rank].wait_event(parts[1 - rank][1])
                    torch.add(parts[0][0], parts[1][0], out=mine)     # rank order: the same bits on both ranks
                return 0
            except Exception as e:      # (an exception must not unwind through the C frames)
                errors.append(repr(e))
                return 13
        return allreduce

    results = [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(0)
            built = syn.build_problem(hip, scene)
            P = built.problem
            P.set_stream(streams[rank].cuda_stream)
            P.set_shard(rank, world)
            P.set_allreduce(make_allreduce(rank))
            o = hip.default_options()
            o.minimizer_progress_to_stdout = 0
            o.max_num_iterations = 50
            o.sync_every = 4
            s = P.solve(o)
            est, ctrl = syn.read_back(built, scene)
            results[rank] = (s, [(i.iteration, i.step_is_successful, i.cost) for i in P.iterations()], est, ctrl, P.comm_info())
            streams[rank].synchronize()
            P.close()
        except Exception as e:
            errors.append("rank %d: %r" % (rank, e))
            meet.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r is not None for r in results)
    single = syn.build_problem(hip, scene)
    o = hip.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 50
    s0 = single.problem.solve(o)
    its0 = [(i.iteration, i.step_is_successful, i.cost) for i in single.problem.iterations()]
    est0, ctrl0 = syn.read_back(single, scene)
    assert calls[0] == calls[1] and calls[0] >= s0.num_iterations
    blocks = 0
    for rank, (s, its, est, ctrl, info) in enumerate(results):
        assert info[0] == rank and info[1] == world and info[3] == scene.num_blocks
        blocks += info[2]
        assert s.termination_type == s0.termination_type == _capi_convergence() and s.num_iterations == s0.num_iterations
        assert [(a, b) for a, b, _ in its] == [(a, b) for a, b, _ in its0]
        for (_, _, c1), (_, _, c0) in zip(its, its0):
            assert abs(c1 - c0) <= 1e-9 * abs(c0)
        assert abs(s.final_cost - s0.final_cost) <= 1e-9 * s0.final_cost
        assert np.abs(ctrl - ctrl0).max() <= 1e-8 * np.abs(ctrl0).max()
        for a, b in zip(est, est0):
            for key in ("intrinsics", "q", "t"):
                assert np.abs(a[key] - b[key]).max() <= 1e-8 * max(1e-3, np.abs(b[key]).max()), key
    assert blocks == scene.num_blocks
    assert np.array_equal(results[0][3], results[1][3])              # replicated solve of one deterministic sum
    assert results[0][1] == results[1][1]


def _capi_convergence():
    from calico_amd import _capi
    return _capi.CONVERGENCE


def _rccl_worker(rank, world, id_bytes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import helpers
    from calico_amd import synthetic as syn
    torch.cuda.set_device(rank)
    torch.zeros(1, device="cuda")
    api = helpers.hip_api()
    scene = _scene()
    built = syn.build_problem(api, scene, device=rank)
    P = built.problem
    P.comm_init_rccl(id_bytes, rank, world)      # one process per GPU; the handle all-reduces natively
    info = P.comm_info()
    s, its = _solve(P, api, 4)
    est, ctrl = syn.read_back(built, scene)
    q.put((rank, s.final_cost, s.num_iterations, s.termination_type, [e["intrinsics"] for e in est], ctrl, its, info))
    P.close()


@pytest.mark.parametrize("world", [2, 4])
def test_native_rccl_ranks_equal_single_rank(world, hip):
    """The production exchange with more than one rank (one process per GPU, ncclAllReduce of the packed normal equations
    over xGMI, commit by copy, zero-filled targets): every rank must walk the single-rank solve's iterations and reach its
    estimates (1e-9: the sum over ranks is associated differently). Needs `world` GPUs in the box."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, torch.cuda.device_count()))
    import multiprocessing as mp
    from calico_amd import _capi, synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = _capi.comm_unique_id(hip)
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, uid, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene = _scene()
    single = syn.build_problem(hip, scene)
    s, its = _solve(single.problem, hip, 4)
    est, ctrl = syn.read_back(single, scene)
    total = 0
    for rank, final_cost, n_it, term, intr, c, rits, info in results:
        assert info[0] == rank and info[1] == world and info[3] == scene.num_blocks
        total += info[2]
        assert term == s.termination_type and n_it == s.num_iterations
        assert abs(final_cost - s.final_cost) <= 1e-9 * s.final_cost
        assert [(a, b) for a, b, _ in rits] == [(a, b) for a, b, _ in its]
        assert np.abs(c - ctrl).max() <= 1e-9 * np.abs(ctrl).max()
        for a, b in zip(intr, est):
            assert np.abs(a - b["intrinsics"]).max() <= 1e-9 * np.abs(b["intrinsics"]).max()
        assert np.array_equal(c, results[0][5])      # replicated solve of a deterministic sum: bit-identical ranks
    assert total == scene.num_blocks                 # every residual block has exactly one owner


def test_comm_info_of_a_single_rank(hip):
    from calico_amd import _capi, synthetic as syn
    scene = _scene()
    b = syn.build_problem(hip, scene)
    assert b.problem.comm_info() == (0, 1, scene.num_blocks, scene.num_blocks)
    b.problem.comm_init_rccl(_capi.comm_unique_id(hip), 0, 1)
    assert b.problem.comm_info() == (0, 1, scene.num_blocks, scene.num_blocks)     # ncclCommCount of a world of one
    b.problem.set_shard(1, 3)
    r, w, nl, nt = b.problem.comm_info()
    assert (r, nt) == (1, scene.num_blocks) and 0 < nl < nt


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` starts N ranks itself; with fewer than N devices it must fail, not report a smaller run."""
    import subprocess
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only %d GPU(s) visible" % (n - 1) in r.stderr and r.stdout.strip() == ""


_LOAD_ORDER_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
from calico_amd import _capi, synthetic as syn
assert "torch" not in sys.modules


def rccl_images():
    seen = set()
    for line in open("/proc/self/maps"):
        path = line.split()[-1]
        if "/" in path and path.rsplit("/", 1)[1].startswith("librccl"):
            seen.add(path)
    return sorted(seen)


hip = _capi.load_hip()
scene = syn.make_scene(2, 1, True, 2, cam_rate=10.0, imu_rate=50.0, duration=3.0, segment_duration=3.0 / 23.9,
                       pixel_noise=0.1, gyro_noise=1e-3, accel_noise=1e-2, robust=True, seed=3)
b = syn.build_problem(hip, scene)
uid = _capi.comm_unique_id(hip)                  # the first calico_comm_* call: the library loads an RCCL of its own
b.problem.comm_init_rccl(uid, 0, 1)
print("IMAGES_BEFORE", len(rccl_images()), flush=True)
sys.stderr.write("MARK\n"); sys.stderr.flush()
import torch                                     # ... and only now the application brings its own
torch.zeros(1, device="cuda")
import torch.distributed                         # (pulls in torch's librccl where torch links it lazily)
b2 = syn.build_problem(hip, scene)
b2.problem.comm_init_rccl(_capi.comm_unique_id(hip), 0, 1)
print("IMAGES_AFTER", len(rccl_images()), flush=True)
o = hip.default_options()
o.minimizer_progress_to_stdout = 0
o.max_num_iterations = 3
s = b2.problem.solve(o)
print("SOLVED", s.num_iterations, flush=True)
import os
os._exit(0)                                      # (two RCCL images: the exit handlers are what this order breaks)
"""


def test_rccl_load_order_is_reported():
    """include/calico_hip.h, LOAD ORDER: a process that makes its first calico_comm_* call BEFORE loading its own RCCL
    (torch imported afterwards) may end up with two librccl images. The library cannot prevent that; calico_comm_init_rccl
    must say so on stderr exactly when it is the case, stay silent otherwise, and the handle must still work."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _LOAD_ORDER_SCRIPT % ROOT], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = dict(line.split()[:2] for line in r.stdout.splitlines() if line.split() and line.split()[0] in ("IMAGES_BEFORE", "IMAGES_AFTER", "SOLVED"))
    assert int(out["IMAGES_BEFORE"]) == 1 and int(out["SOLVED"]) >= 1
    before, _, after = r.stderr.partition("MARK\n")
    assert "two librccl images" not in before
    assert ("two librccl images" in after) == (int(out["IMAGES_AFTER"]) > 1)
