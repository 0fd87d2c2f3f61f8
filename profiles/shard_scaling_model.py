#!/usr/bin/env python
"""What one rank of an N-GPU run would spend per LM iteration, measured on ONE GPU: the rank-local evaluation chain
(Jacobian kernel, cell expansion, gather) of shard 0 of N for N = 1, 2, 4, 8 (calico_problem_set_shard + an exchange
callback that does nothing: the numbers of the normal equations are wrong, the time of the rank-local kernels is
right), next to the rest of the iteration (linear solve, per-solve overheads: wall time per iteration of plain solves
minus the evaluation chain), which every rank runs in full and which does not depend on N.

    python profiles/shard_scaling_model.py [config ...]  > gpurun_out/r03/r03_shard_scaling_model.json

HIP-event brackets around each phase (an event pair costs ~6 us of stream time: calibrated and subtracted). The
all-reduce itself cannot be measured on one GPU; the table is what the scaling curve can at best look like before its
latency is added."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    import torch
    torch.zeros(1, device="cuda")
    from calico_amd import _capi, synthetic as syn
    api = _capi.load_hip()
    out = {}
    for cfg in [int(a) for a in sys.argv[1:]] or [3, 4]:
        scene = syn.config_scene(cfg)
        rows = []
        # one LM iteration of a plain single-rank solve, wall clock over whole solves with no event brackets on the stream
        import time
        o = api.default_options()
        o.minimizer_progress_to_stdout = 0
        b = syn.build_problem(api, scene)
        b.problem.solve(o)
        b.problem.close()
        it_n, it_t = 0, 0.0
        for _ in range(5):
            b = syn.build_problem(api, scene)
            b.problem.finalize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sres = b.problem.solve(o)
            torch.cuda.synchronize()
            it_t += time.perf_counter() - t0
            it_n += sres.num_iterations
            b.problem.close()
        iter_us = 1e6 * it_t / max(1, it_n)
        bracket = None
        for world in (1, 2, 4, 8):
            b = syn.build_problem(api, scene)
            if world > 1:
                b.problem.set_shard(0, world)
                b.problem.set_allreduce(lambda ctx, buf, n, strm: 0)
            b.problem.set_phase_timing(0x3f)
            o1 = api.default_options()
            o1.minimizer_progress_to_stdout = 0
            o1.max_num_iterations = 1
            b.problem.solve(o1)                      # bracket calibration (phase 5) + warm-up
            for _ in range(30):
                try:
                    b.problem.evaluate(want_jtj=False)
                except _capi.CalicoError:
                    pass
            ph = [b.problem.phase_time(i | 0x100) for i in range(2)]
            if bracket is None:
                cal = b.problem.phase_time(5)
                bracket = cal[0] / max(1, cal[1]) - 0.002
            _, _, nl, nt = b.problem.comm_info()
            ev = 1e3 * (ph[0][0] / max(1, ph[0][1]) - bracket)
            ga = 1e3 * (ph[1][0] / max(1, ph[1][1]) - bracket)
            rows.append({"world": world, "blocks_on_rank_0": nl, "blocks_total": nt, "jacobian_kernel_us": round(ev, 2),
                         "expand_plus_gather_us": round(ga, 2)})
            b.problem.close()
        # everything that is not the evaluation chain (linear solve, per-solve overheads) is the same on every rank
        rest = iter_us - rows[0]["jacobian_kernel_us"] - rows[0]["expand_plus_gather_us"]
        for r in rows:
            r["replicated_rest_us"] = round(rest, 2)
            r["iteration_us_before_the_all_reduce"] = round(rest + r["jacobian_kernel_us"] + r["expand_plus_gather_us"], 2)
        base = rows[0]["iteration_us_before_the_all_reduce"]
        for r in rows:
            r["speedup_bound_without_all_reduce"] = round(base / r["iteration_us_before_the_all_reduce"], 3)
        out["configs[%d]" % cfg] = {"residual_blocks": scene.num_blocks, "control_points": len(scene.ctrl),
                                    "single_rank_iteration_us": round(iter_us, 2), "per_rank": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
