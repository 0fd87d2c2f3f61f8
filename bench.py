#!/usr/bin/env python
"""Benchmark of the hot path: Levenberg-Marquardt iterations per second on the
4-camera + IMU, ~100k-observation synthetic problem (BASELINE.json configs[3]).

A "step" is one LM iteration of calico_solve (linear solve + candidate cost
evaluation + accept/reject, and a residual/Jacobian/JtJ evaluation whenever the
step is accepted) with all observations resident in HBM. The timed region runs
whole solves from the perturbed initial guess (reference default tolerances)
until exactly K iterations have been made; each solve's initial evaluation
(iteration 0) is inside the timed region and not counted as a step.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_jacobian_launch(scene):
    """SURVEY.md §8(d) per-block figure restricted to what ONE launch of the fused
    residual/Jacobian/JtJ kernel stands for: read the observation (40 B), write the
    residual (8d), write the Jacobian (8dc), read both back for assembly (8dc + 8d).
    d = residual dim, c = active tangent columns of the block."""
    from calico_amd import _capi
    total = 0
    k6 = 6 * scene.order
    for s in scene.sensors:
        c = k6
        if s.enable_intrinsics:
            c += len(s.intrinsics)
        if s.enable_extrinsics:
            c += 3 + (0 if s.kind == _capi.SENSOR_GYROSCOPE else 3)
        if s.enable_latency:
            c += 1
        d = s.dim
        total += s.n * (40 + 16 * d + 16 * d * c)
    return total


def algorithmic_bytes_per_iteration(scene):
    """SURVEY.md 8(d) per LM iteration: the Jacobian evaluation above plus the cost-only evaluation of the candidate
    (observation read + residual written once more): 2·obs + 3·8d + 2·8dc per block."""
    total = algorithmic_bytes_per_jacobian_launch(scene)
    for s in scene.sensors:
        total += s.n * (40 + 8 * s.dim)
    return total


def host_cpu_info():
    """Model string, physical cores and sockets of the host from /proc/cpuinfo, and the CPUs this process may run on."""
    model, pairs, sockets = None, set(), set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
                sockets.add(v)
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                pairs.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    return {"model": model or "unknown", "physical_cores": len(pairs) or None, "sockets": len(sockets) or None,
            "cpus_allowed": allowed}


def cpu_baseline(scene, min_seconds=10.0):
    """The CPU oracle (restatement of the reference's Ceres path: dual-number autodiff in 4-wide passes, dense normal
    equations, dense Cholesky, std::thread pool honouring num_threads) on the host cores of this box, as SURVEY.md 8(d) /
    BASELINE.md describe the baseline: num_threads in {1, 4 (the demos' setting), all cores}, seconds per iteration split
    into evaluate / assemble / linear solve, iterations to convergence. A bounded sample of the same workload -- whole
    solves from the same perturbed start; the headline `value` is the all-cores figure. Reported, never shipped."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from calico_amd import synthetic as syn
    api = helpers.oracle_api()
    cores = os.cpu_count() or 1
    all_threads = max(1, min(cores, 64))
    host = host_cpu_info()

    def run(threads, iters, budget_s, max_solves):
        o = api.default_options()
        o.minimizer_progress_to_stdout = 0
        o.max_num_iterations = iters
        o.num_threads = threads
        nit = solves = 0
        dt = 0.0
        part = [0.0, 0.0, 0.0]
        last = None
        while (solves == 0 or dt < budget_s) and solves < max_solves:
            built = syn.build_problem(api, scene)      # fresh problem = same perturbed start
            t = time.time()
            last = built.problem.solve(o)
            dt += time.time() - t
            t3 = (C.c_double * 3)()
            api.lib.oracle_get_solve_timing(built.problem.h, t3)
            part = [part[i] + t3[i] for i in range(3)]
            nit += max(1, last.num_iterations)
            solves += 1
        return {"num_threads": threads, "iterations_per_s": nit / dt, "s_per_iteration": dt / nit,
                "s_per_iteration_evaluate": part[0] / nit, "s_per_iteration_assemble": part[1] / nit,
                "s_per_iteration_linear_solve": part[2] / nit, "iterations": nit, "solves": solves, "seconds": dt}, last

    sweep = []
    for th, iters in ((1, 3), (4, 6)):
        if th < all_threads:
            sweep.append(run(th, iters, 0.0, 1)[0])
    full, last = run(all_threads, 50, min_seconds, 20)
    sweep.append(full)
    # "cores" = the threads the all-cores leg actually ran on (the oracle's pool is capped at 64); what the box has is
    # stated next to it: logical CPUs, physical cores, sockets and the model string (north_star: "core count stated")
    return {"value": full["iterations_per_s"], "unit": "LM iterations/s", "cores": all_threads, "kind": "port",
            "num_threads": all_threads, "host_logical_cpus": cores, "host_physical_cores": host["physical_cores"],
            "host_sockets": host["sockets"], "host_cpu_model": host["model"],
            "host_cpus_allowed": host["cpus_allowed"],
            "iterations_to_convergence": int(last.num_iterations), "termination": last.message.decode(),
            "by_num_threads": sweep,
            "sample": "%d LM iterations in %d solves to convergence (+ initial evaluations) of the same %d-block problem on %d threads, "
                      "%.1f s of solver time; 1- and 4-thread legs: the first 3 / 6 iterations of the same solve"
                      % (full["iterations"], full["solves"], scene.num_blocks, all_threads, full["seconds"])}


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def visible_gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run would set them) and wait for all of them; rank 0 prints the JSON line. Fails loudly when the
    node has fewer than N devices -- a run that silently used fewer GPUs than asked for would report the wrong n_gpus."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < n:
        sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this node -- refusing to run on fewer devices than asked for\n" % (n, have))
        return 3
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        while procs and rc == 0:
            for pr in list(procs):
                code = pr.poll()
                if code is None:
                    continue
                procs.remove(pr)
                if code != 0:
                    rc = code
            time.sleep(0.05)
    finally:
        for pr in procs:       # a rank failed: stop the ones we started (exact PIDs)
            pr.kill()
    return rc


def _profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except Exception:
        return None


def dominant_kernel_from_profiles():
    """(kernel name, share of the kernel time of ONE iteration, average working launch in us, file) of the per-iteration
    kernel with the longest launch in the newest committed profiles/rNN_kernel_stats.csv. Per-iteration kernels: the ones
    launched (about) once per LM iteration -- working calls within a factor two of the most-launched solver kernel --, so
    that set-up kernels (gather lists, seeds: a few long launches per handle) and the evaluation's extra launches at the
    start of every solve (more CALLS than the solver kernels, hence the largest TOTAL duration) do not decide it: what
    bounds an iteration is the longest launch in its chain."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats.csv")))
    if not files:
        return None
    rows = [r for r in csv.DictReader(open(files[-1])) if "cal" in r["Name"]]
    if not rows:
        return None
    calls = lambda r: float(r.get("WorkingCalls") or r["Calls"])
    most = max(calls(r) for r in rows)
    per_iteration = [r for r in rows if calls(r) >= 0.5 * most]
    top = max(per_iteration, key=lambda r: float(r["AverageWorkingNs"]))
    total = sum(float(r["AverageWorkingNs"]) for r in per_iteration)
    return top["Name"], float(top["AverageWorkingNs"]) / total, float(top["AverageWorkingNs"]) / 1e3, os.path.basename(files[-1])


def expected_scaling_bound(config, world):
    """Upper bound on the speed-up of `world` ranks (all-reduce free), measured on one GPU by profiles/shard_scaling_model.py:
    every rank solves the whole linear system, only the evaluation shrinks with the shard."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_shard_scaling_model.json")))
    if not files:
        return None
    try:
        rows = json.load(open(files[-1])).get("configs[%d]" % config, {}).get("per_rank", [])
        for r in rows:
            if r.get("world") == world:
                return {"speedup_over_one_rank_at_most": r["speedup_bound_without_all_reduce"], "source": "profiles/" + os.path.basename(files[-1]),
                        "note": "every rank solves the whole linear system (DESIGN.md 8): only the evaluation shrinks with the shard, and the "
                                "all-reduce of R comes on top"}
    except Exception:
        pass
    return None


def roofline_object(scene, world, args, ms_per_step, iter_bytes, dense_ms, dense_ms_raw, phase_n, jac_ms, jac_ms_raw, jac, bracket_ms,
                    n_skipped, alg_bytes, achieved, wu_ms, wu_n):
    not_this_run = "profiles/ (committed rocprofv3 passes on the default workload, one rank; NOT measured in this run)"
    use_profiles = args.config == 3 and world == 1
    traffic_js = _profile_json("hbm_traffic.json") if use_profiles else None
    fp64_js = _profile_json("fp64_utilisation.json") if use_profiles else None

    def counters(match):
        out = {"source": not_this_run}
        if traffic_js:
            for k, v in (traffic_js.get("by_kernel") or {}).items():
                if match in k:
                    out["hbm_bytes_per_launch"] = v
                    out["hbm_kernel"] = k
                    break
        if fp64_js:
            for k, v in (fp64_js.get("kernels") or {}).items():
                if match in k:
                    out["fp64_frac_of_78.6_tflops"] = v.get("fp64_frac")
                    out["mfma_util"] = v.get("mfma_util")
                    out["profiled_launch_us"] = v.get("us")
                    break
        return out

    dom = dominant_kernel_from_profiles()
    dense_c = counters("dense_back_kernel")
    eval_c = counters("eval_cells_kernel")
    if "hbm_bytes_per_launch" not in eval_c and "mfma_util" not in eval_c:
        eval_c = counters("eval_jacobian_kernel")        # (profiles of a plan without cell workgroups)
    lin_ms = max(0.0, wu_ms[2] / max(1, wu_n[2]) - bracket_ms)      # one linear solve (its launches back to back), warmup solves
    dominant = {
        "kernel": "dense_back_kernel (dense reduced solve of the calibration + root block, first back-substitution launch behind an in-launch hand-off)"
                  if phase_n[8] else "n/a (no reduced-system launch bracketed)",
        "by_profile": {"kernel": dom[0], "share_of_kernel_time": dom[1], "avg_working_launch_us": dom[2], "file": "profiles/" + dom[3]} if dom else None,
        "avg_launch_ms": dense_ms if phase_n[8] else None, "launches_bracketed": phase_n[8],
        "avg_launch_ms_with_event_bracket": dense_ms_raw if phase_n[8] else None,
        "share_of_step": (dense_ms / ms_per_step) if phase_n[8] else None,
        "limited_by": "latency: one workgroup factors the reduced system block by block (a chain of dependent 4-column steps on one wave), "
                      "six more wait for its solution; neither HBM nor FP64 throughput is the bound (see counters)",
        "measured_frac_of_hbm_peak": (dense_c["hbm_bytes_per_launch"] / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                     if phase_n[8] and "hbm_bytes_per_launch" in dense_c else None,
        "counters": dense_c,
    }
    evaluation = {
        "kernel": "eval_cells_kernel (fused residual + analytic Jacobian + JtJ blocks of whole cells: two-wave workgroups, a camera cell's "
                  "frames expand their block out of LDS; eval_jacobian_kernel + expand_cells_kernel where a plan has no cell workgroups)",
        "avg_launch_ms": jac_ms, "launches": jac, "launches_bracketed": phase_n[7],
        "avg_launch_ms_with_event_bracket": jac_ms_raw, "bracketed_launches_that_exited_early": n_skipped,
        "share_of_step": jac_ms / ms_per_step,
        "algorithmic_bytes_per_launch": alg_bytes,
        "measured_frac_of_hbm_peak": (eval_c["hbm_bytes_per_launch"] / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if "hbm_bytes_per_launch" in eval_c else None,
        "fused_paper_ratio": achieved / HBM_PEAK_GBS,
        "fused_paper_ratio_note": "algorithmic bytes of the UNFUSED data flow of one Jacobian launch over its time: a paper figure, "
                                  "the kernel does not move those bytes",
        "counters": eval_c,
    }
    # The dominant kernel of THIS run: the longer of the two launches timed live with HIP events (the reduced system's launch
    # and the evaluation launch; the tree levels are shorter than either at every configuration measured) -- the committed
    # profile only says which kernels are candidates (dominant_kernel_from_profiles ranks by average launch per iteration).
    eval_dominates = (not phase_n[8]) or jac_ms > dense_ms
    live = evaluation if eval_dominates else dominant
    live_c = eval_c if eval_dominates else dense_c
    dominant["is_dominant_in_this_run"] = not eval_dominates
    evaluation["is_dominant_in_this_run"] = eval_dominates
    return {
        "bound": "hbm", "limited_by": "latency (dependent FP64 chains of single waves and five kernel boundaries per iteration; DESIGN.md 4)",
        "achieved": iter_bytes / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": iter_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes_per_iteration": iter_bytes,
        # `traffic`: HBM bytes per launch of the dominant kernel (PMC, per the guide's corrections); null when there is no committed pass for this run's shape
        "traffic": live_c.get("hbm_bytes_per_launch"), "traffic_source": not_this_run,
        "kernel": live["kernel"], "kernel_avg_launch_ms": live["avg_launch_ms"], "event_bracket_ms": bracket_ms,
        "linear_solve_share_of_step": lin_ms / ms_per_step if wu_n[2] else None,
        "linear_solve_ms": lin_ms if wu_n[2] else None,
        "dominant_kernel": dominant,
        "evaluation_kernel": evaluation,
        "mfma_utilisation": {"kernel": "eval_cells_kernel" if eval_dominates else "dense_back_kernel", "mfma_util": live_c.get("mfma_util"), "fp64_frac": live_c.get("fp64_frac_of_78.6_tflops"),
                             "source": not_this_run,
                             "definition": "SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) / (kernel duration x 2.4 GHz x 1024 SIMDs); fp64_frac: FP64 flops / duration / 78.6 TFLOP/s"},
        "fp64_by_kernel": ({"source": not_this_run, "kernels": fp64_js.get("kernels")} if fp64_js else None),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config index (3 = north star)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-every", type=int, default=8)
    ap.add_argument("--force-collective", action="store_true",
                    help="exercise the sharding + all-reduce path even with one rank (validation)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed samples of --steps iterations each (default: 25 when --steps <= 50, else 5); the median is reported")
    ap.add_argument("--min-seconds", type=float, default=10.5,
                    help="without --repeats: as many samples as it takes to keep the GPU busy this long")
    ap.add_argument("--poor-start", type=float, default=0.0,
                    help="move the start away from the truth: focal lengths x (1 + F/100), translations + F cm, control "
                         "points + F mrad / F mm of noise (context runs: many rejected steps); 0 = the reference test's perturbation")
    ap.add_argument("--tagging-passes", type=int, default=0,
                    help="outlier tagging loop (configs[4]): solve, tag |r| > 3 on the device, re-solve, N times; reported, untimed")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match the launcher's WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))

    # stdout carries exactly ONE JSON line: libraries that print to the C-level stdout (RCCL's version banner at
    # communicator creation, kernel debug prints) are sent to stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    from calico_amd import _capi, synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcalico_hip.so is the only backend")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d of %d has no device (%d GPU(s) visible)" % (rank, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    collective = world > 1 or args.force_collective
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    api = _capi.load_hip()
    scene = syn.config_scene(args.config)
    built = syn.build_problem(api, scene, device=local_rank)
    P = built.problem
    if collective:
        # native exchange: the handle owns an RCCL communicator and issues ncclAllReduce itself on its own stream (no Python
        # in the loop). torch.distributed only carries the 128-byte id from rank 0 to the others, and the timing barrier.
        P.set_stream(torch.cuda.current_stream().cuda_stream)
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt = torch.tensor(list(_capi.comm_unique_id(api)), dtype=torch.uint8, device="cuda")
        dist.broadcast(idt, src=0)
        P.comm_init_rccl(bytes(idt.cpu().tolist()), rank, world)
    # what the exchange really spans (ncclCommCount) and what this rank evaluates
    comm_rank, comm_world, local_blocks, total_blocks = P.comm_info()
    if comm_world != world or comm_rank != rank:
        raise SystemExit("bench.py: the communicator spans %d ranks (this is rank %d), the launcher said %d (rank %d)" % (comm_world, comm_rank, world, rank))
    blocks_per_rank = [local_blocks]
    if dist is not None and world > 1:
        t = torch.zeros(world, dtype=torch.int64, device="cuda")
        t[rank] = local_blocks
        dist.all_reduce(t)
        blocks_per_rank = [int(v) for v in t.cpu().tolist()]
        if sum(blocks_per_rank) != total_blocks:
            raise SystemExit("bench.py: the shards hold %d residual blocks, the problem has %d" % (sum(blocks_per_rank), total_blocks))

    # initial values of every free block, to restart solves inside the timed region
    init = [(int(b), scene.ctrl[i].copy()) for i, b in enumerate(built.ctrl_blocks)]
    for s, sb in zip(scene.sensors, built.sensor_blocks):
        init += [(sb["intrinsics"], s.intrinsics.copy()), (sb["t"], s.t.copy()), (sb["q"], s.q.copy()),
                 (sb["latency"], np.array([s.latency]))]

    if args.poor_start > 0:
        rng = np.random.default_rng(1234)
        F = args.poor_start
        init = [(int(b), scene.ctrl[i] + 1e-3 * F * rng.standard_normal(6))
                for i, b in enumerate(built.ctrl_blocks)]
        for s_, sb in zip(scene.sensors, built.sensor_blocks):
            intr = s_.intrinsics.copy()
            if s_.kind == _capi.SENSOR_CAMERA:
                intr[0] *= 1.0 + F / 100.0
            init += [(sb["intrinsics"], intr), (sb["t"], s_.t + 0.01 * F * rng.uniform(-1, 1, 3)), (sb["q"], s_.q.copy()),
                     (sb["latency"], np.array([s_.latency]))]
    init_ids = np.array([b for b, _ in init], np.int32)
    init_vals = np.concatenate([np.asarray(v, float).ravel() for _, v in init])

    # the restart of a solve through the raw ABI call with pointers made once: the host time between two solves is part
    # of the timed region, and numpy -> ctypes conversions cost more than the call itself
    _ids_p = init_ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    _vals_p = init_vals.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def reset():
        if api.set_param_blocks(P.h, int(init_ids.size), _ids_p, _vals_p) != 0:
            raise RuntimeError("calico_set_param_blocks failed")

    opts = api.default_options()
    opts.minimizer_progress_to_stdout = 0
    opts.sync_every = args.sync_every  # LM iterations enqueued per host round trip (single GPU)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_solves(n):
        """n LM iterations over whole solves; returns the counters (phase times: read_phases, after the closing barrier)."""
        done = jac = cost = solves = 0
        last = None
        while done < n:
            reset()
            opts.max_num_iterations = min(50, n - done)
            s = P.solve(opts)
            if s.num_iterations <= 0:
                raise RuntimeError("solve made no progress: %s" % s.message.decode())
            # a timed iteration only counts if the optimiser is really optimising: a solve given the full budget from the
            # perturbed start must converge, and every solve must take successful steps that lower the cost
            # (convergence within Ceres' default 50 iterations is demanded of the benchmark workload; the context
            #  configurations -- 50 Hz knots, the EuRoC shape -- are weakly constrained and take longer)
            if not (s.num_successful_steps > 0 and s.final_cost < s.initial_cost) or \
                    (args.config <= 4 and args.poor_start == 0 and opts.max_num_iterations >= 50 and s.termination_type != _capi.CONVERGENCE):
                raise RuntimeError("solve did not behave (%g -> %g, %d successful steps, termination %d): %s" % (
                    s.initial_cost, s.final_cost, s.num_successful_steps, s.termination_type, s.message.decode()))
            done += s.num_iterations
            jac += s.num_jacobian_evaluations
            cost += s.num_cost_evaluations
            solves += 1
            last = s
        return done, jac, cost, solves, last

    def read_phases(phases=range(6)):
        """HIP-event phase times accumulated since the last set_phase_timing (waits for the stream: a solve returns as soon
        as the device reports its end, the early-exit kernels of the iterations enqueued ahead drain afterwards)."""
        phase_ms = [0.0] * 9
        phase_n = [0] * 9
        for i in phases:
            phase_ms[i], phase_n[i] = P.phase_time(i)
        phase_ms[7], phase_n[7] = P.phase_time(0 | 0x100)   # Jacobian launches that did work (not the early exits after termination)
        phase_ms[8], phase_n[8] = P.phase_time(6 | 0x100)   # the same for the launch that solves the reduced system
        return phase_ms, phase_n

    # Optimize() end to end on a fresh handle: the reference rebuilds its ceres::Problem on every call
    # (batch_optimizer.cpp:57-70), so flattening + upload (setup) and the copy-back of estimates and residuals
    # (writeback) belong to the path; reported next to the per-iteration figure, not inside it
    setup_ms = writeback_ms = setup_add_ms = None
    unseen_samples = None
    setup_known = None
    if rank == 0 and world == 1:
        def one_setup():
            t = time.perf_counter()
            fresh = syn.build_problem(api, scene, device=local_rank)      # add_* calls: host-side flattening of the sensors
            t_add = time.perf_counter()
            fresh.problem.finalize()                                      # plan (built or from the cache), workspace, value uploads
            torch.cuda.synchronize()
            return fresh, 1e3 * (time.perf_counter() - t), 1e3 * (t_add - t)
        # (a) a structure the library has not seen (plan cache emptied): cells, work items, gather lists, elimination plan, H2D.
        # First in a cold process (first allocations of the host tables, first hipMalloc of the sizes), then, after (b), on a
        # structural variant in the warm process.
        api.plan_cache_clear()
        fresh, setup_cold_ms, setup_cold_add_ms = one_setup()
        o1 = api.default_options()
        o1.minimizer_progress_to_stdout = 0
        o1.max_num_iterations = 3
        fresh.problem.solve(o1)
        t = time.perf_counter()
        syn.read_back(fresh, scene)                                   # every estimate back into host objects
        for sid, sp in zip(fresh.sensor_ids, scene.sensors):
            fresh.problem.residuals(sid, sp.n, sp.dim)                # Sensor::UpdateResiduals
        writeback_ms = 1e3 * (time.perf_counter() - t)
        fresh.problem.close()
        # (b) the reference's pattern -- the same problem rebuilt for the next Optimize(): plan and workspace come from the cache
        known = []
        for _ in range(5):
            fresh, t_all, t_add = one_setup()
            known.append((t_all, t_add))
            fresh.problem.close()
        known.sort()
        setup_known = known[len(known) // 2]
        # (a') unseen structures in the warm process: the same scene with one camera observation left out (another one
        # each time: every variant is planned afresh), median of 5 after two untimed ones
        import copy
        unseen = []
        for drop in range(-2, 5):         # (two untimed variants first: the first new structure of a process also pays first-time
                                          #  device allocations at these sizes, 16-30 ms -- that is `setup_ms_cold_process`'s business)
            var = copy.copy(scene)
            var.sensors = list(scene.sensors)
            cam = copy.copy(scene.sensors[0])
            keep = np.ones(cam.n, bool)
            keep[17 + drop + 2] = False
            cam.meas, cam.stamps, cam.point_idx = cam.meas[keep], cam.stamps[keep], cam.point_idx[keep]
            var.sensors[0] = cam
            h0, m0, _ = _capi.plan_cache_stats(api)
            t = time.perf_counter()
            fresh = syn.build_problem(api, var, device=local_rank)
            t_add = time.perf_counter()
            fresh.problem.finalize()
            torch.cuda.synchronize()
            if drop >= 0:
                unseen.append((1e3 * (time.perf_counter() - t), 1e3 * (t_add - t)))
            assert _capi.plan_cache_stats(api)[1] == m0 + 1          # really planned afresh
            fresh.problem.close()
        unseen_samples = [round(t_all, 3) for t_all, _ in unseen]      # (in measurement order: reported beside the median)
        unseen.sort()
        setup_ms, setup_add_ms = unseen[len(unseen) // 2]

    # warmup: all phases bracketed by HIP events -> per-phase breakdown (reported, untimed)
    P.set_phase_timing(0x3f)   # all phases + the bracket calibration (phase 5)
    timed_solves(max(1, args.warmup))
    wu_ms, wu_n = read_phases()
    # timed region: only the Jacobian kernel (phase 0) and the reduced-system launch (phase 6: the longest kernel of an
    # iteration) carry events, and only every 64th of their launches (one launch of each per sample; round 4: every 16th,
    # two per sample and 1.5 % of the sample's time) -- an event pair costs ~6 us of stream time on either side of the kernel. The K-step sample is a few milliseconds long, so it is
    # repeated and the MEDIAN sample is the one reported (box-to-box and run-to-run spread is several per cent).
    # By default the samples add up to >= 8 s of contiguous GPU work (9 s are aimed at: the first, estimating sample runs slower) (a 20-iteration sample is ~3 ms: the driver's 5-second utilisation
    # sampler would otherwise never see the device busy); --repeats N fixes the count.
    if args.repeats > 0:
        repeats = args.repeats
    else:
        barrier()
        t0 = time.perf_counter()
        timed_solves(args.steps)
        barrier()
        est = time.perf_counter() - t0
        repeats = int(min(5000, max(25 if args.steps <= 50 else 5, np.ceil(args.min_seconds / max(est, 1e-4)))))
        if dist is not None:     # every rank must run the same number of samples
            t = torch.tensor([repeats], device="cuda", dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            repeats = int(t.item())
    samples = []
    for _ in range(repeats):
        P.set_phase_timing(0x41 | (64 << 8))             # (restarts the accumulated phase times; every 64th launch: the first of each sample)
        barrier()
        t0 = time.perf_counter()
        rec = timed_solves(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        rec = rec + read_phases((0, 6))                  # only the Jacobian kernel and the reduced-system launch carry events in the timed region
        if dist is not None:
            t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        samples.append((elapsed / rec[0], elapsed, rec))
    samples.sort(key=lambda z: z[0])
    _, elapsed, (done, jac, cost, solves, last, phase_ms, phase_n) = samples[len(samples) // 2]

    tagging = None
    if args.tagging_passes > 0 and rank == 0 and world == 1:
        # the demos' outlier loop on the device: Optimize -> residuals -> tag |r| > 3 -> Optimize, untimed
        reset()
        opts.max_num_iterations = 50
        tagging = []
        for _ in range(args.tagging_passes):
            t = time.perf_counter()
            sres = P.solve(opts)
            marked = sum(P.mark_outliers(sid, 3.0) for sid, sp in zip(built.sensor_ids, scene.sensors) if sp.kind == _capi.SENSOR_CAMERA)
            tagging.append({"iterations": sres.num_iterations, "final_cost": sres.final_cost, "tagged": int(marked),
                            "ms": 1e3 * (time.perf_counter() - t)})

    if rank == 0:
        n_blocks = scene.num_blocks
        # HIP-event time of the sampled Jacobian launches that did work (a launch enqueued ahead of a solve that has
        # terminated exits at once), minus what the same event bracket measures around a ~2 us kernel
        bracket_ms = max(0.0, wu_ms[5] / max(1, wu_n[5]) - 0.002)     # calibrated during the warmup solves
        n_skipped = max(0, phase_n[0] - phase_n[7])
        jac_ms_raw = phase_ms[7] / max(1, phase_n[7])
        jac_ms = max(1e-6, jac_ms_raw - bracket_ms)
        dense_ms_raw = phase_ms[8] / max(1, phase_n[8])
        dense_ms = max(1e-6, dense_ms_raw - bracket_ms)
        alg_bytes = algorithmic_bytes_per_jacobian_launch(scene) / world
        achieved = alg_bytes / (jac_ms * 1e-3) / 1e9 if jac_ms > 0 else 0.0
        ms_per_step = 1e3 * elapsed / done
        iter_bytes = algorithmic_bytes_per_iteration(scene)      # whole job: all ranks together
        out = {
            "metric": "LM iterations/sec on the 4-cam+IMU ~100k-observation problem",
            "value": done / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": comm_world,
            "steps": done,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "repeats": repeats,
            "timed_seconds_total": sum(z[1] for z in samples),
            # (sorted samples; the median is the one reported) quantiles of all of them + the 25 around the median
            "ms_per_step_quantiles": {q: round(1e3 * samples[min(len(samples) - 1, int(q * len(samples) / 100))][0], 5) for q in (0, 10, 25, 50, 75, 90, 99)},
            "ms_per_step_samples": [round(1e3 * z[0], 5) for z in samples[max(0, len(samples) // 2 - 12):len(samples) // 2 + 13]],
            "config": {
                "workload": "%s: %d cameras + %d gyro + %d accel, %d residual blocks (%d scalar residuals), "
                            "%d control points, robust kernels %s" % (
                                ("BASELINE.json configs[%d]" % args.config) if args.config <= 4 else
                                {5: "context shape (not a BASELINE config): configs[3] with 50 Hz knots",
                                 6: "context shape (not a BASELINE config): one camera + IMU, 1453 control points (the notebook run's shape)"}.get(args.config, "context shape %d" % args.config),
                                sum(1 for s in scene.sensors if s.kind == 0),
                                sum(1 for s in scene.sensors if s.kind == _capi.SENSOR_GYROSCOPE),
                                sum(1 for s in scene.sensors if s.kind == _capi.SENSOR_ACCELEROMETER), n_blocks,
                                sum(s.n * s.dim for s in scene.sensors), len(scene.ctrl),
                                "on" if any(s.loss for s in scene.sensors) else "off"),
                "residual_blocks": n_blocks,
                "residual_blocks_per_rank": blocks_per_rank,
                "effective_parameters": last.num_effective_parameters_reduced,
                "solves_in_timed_region": solves,
                "jacobian_evaluations": jac,
                "cost_evaluations": cost,
                # one fused pass per iteration evaluates residuals, cost and Jacobian of every block (speculative
                # evaluation at the candidate point): the blocks are counted once per pass
                "residual_blocks_evaluated_per_s": n_blocks * max(jac, cost) / elapsed,
                "parallelism": "obs-shard x%d + native RCCL all-reduce(JtJ,Jtr,cost)" % world if world > 1 else "single GPU",
                "host_loop": "non-blocking: device-published progress; iteration i + 1 enqueued when the Jacobian launch of iteration i reports that its control stage will not end the solve (CALICO_PREDICT_END=0: always one iteration ahead)" if world == 1 and not args.force_collective else "batches of %d iterations per host read-back" % args.sync_every,
                "poor_start": args.poor_start,
                "successful_steps_last_solve": last.num_successful_steps, "unsuccessful_steps_last_solve": last.num_unsuccessful_steps,
                "linear_solver": os.environ.get("CALICO_SOLVER", "tree (block cyclic reduction over 5-control-point superblocks)"),
                # fresh handle, structure not seen before (planned afresh), warm process: add_* calls + plan + workspace + uploads
                "setup_ms": setup_ms,
                "setup_ms_samples": unseen_samples if setup_ms else None,
                "setup_ms_cold_process": setup_cold_ms, "setup_ms_cold_process_add_calls": setup_cold_add_ms,
                "setup_ms_add_calls": setup_add_ms,        # of which: the add_* calls through the C ABI (python + ctypes here)
                "setup_ms_finalize": (setup_ms - setup_add_ms) if setup_ms else None,
                # fresh handle, structure seen before (the reference rebuilds the same problem per Optimize()): median of 5
                "setup_ms_known_structure": setup_known[0] if setup_known else None,
                "setup_ms_known_structure_add_calls": setup_known[1] if setup_known else None,
                "setup_ms_known_structure_finalize": (setup_known[0] - setup_known[1]) if setup_known else None,
                "writeback_ms": writeback_ms,
                "setup_in_iterations": (setup_ms / ms_per_step) if setup_ms else None,
                "phase_ms_per_launch_warmup": {
                    "jacobian_eval": wu_ms[0] / max(1, wu_n[0]),
                    "gather": wu_ms[1] / max(1, wu_n[1]),
                    "linear_solve": wu_ms[2] / max(1, wu_n[2]),
                    "cost_eval": 0.0,
                    "control": wu_ms[4] / max(1, wu_n[4]),
                },
            },
            # `frac` is SURVEY.md 8(d)'s own definition: the ALGORITHMIC bytes of one LM iteration (per block: observation read
            # twice, residual written, Jacobian written and read back for assembly, residual of the cost-only pass) over the
            # measured time of an iteration, against the HBM peak. The fused kernels keep the Jacobian on chip, so the bytes
            # the hardware really moves are far fewer (PMC figures under `dominant_kernel` / `evaluation_kernel`).
            # `dominant_kernel` is the kernel with the largest share of the step in profiles/<round>_kernel_stats.csv -- the
            # launch that solves the reduced system; it and the Jacobian kernel are timed live (HIP events on the library's
            # stream); counters (`traffic`, FP64 / MFMA utilisation) come from the committed rocprofv3 passes, not from this run.
            "roofline": roofline_object(scene, world, args, ms_per_step, iter_bytes, dense_ms, dense_ms_raw, phase_n, jac_ms, jac_ms_raw,
                                        jac, bracket_ms, n_skipped, alg_bytes, achieved, wu_ms, wu_n),
        }
        if world > 1:
            bound = expected_scaling_bound(args.config, world)
            if bound is not None:
                out["expected_strong_scaling_bound"] = bound
        if tagging is not None:
            out["config"]["outlier_tagging_passes"] = tagging
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
