#!/usr/bin/env python
"""Run-to-run determinism of the HIP path: the same solve repeated must give bit-identical iterates
(no atomics, fixed-order reductions). A data race shows up here long before it shows up in a tolerance test.

  python profiles/microbench/determinism_check.py [--config 3] [--repeats 6]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from calico_amd import _capi, synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=6)
    args = ap.parse_args()
    api = _capi.load_hip()
    scene = syn.config_scene(args.config)
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
    ref = None
    bad = 0
    for r in range(args.repeats):
        P.set_param_blocks(ids, vals)
        s = P.solve(o)
        its = P.iterations()
        sig = (s.num_iterations, s.termination_type, tuple(float(i.cost) for i in its))
        x = np.concatenate([np.asarray(P.get_param_block(int(b), n), float).ravel() for b, n in zip(ids, sizes)])
        if ref is None:
            ref = (sig, x)
            print("run 0: %d iterations, termination %d, final cost %.17g" % (s.num_iterations, s.termination_type, s.final_cost))
        else:
            same = sig == ref[0] and np.array_equal(x, ref[1])
            if not same:
                bad += 1
                print("run %d DIFFERS: %d iterations, final cost %.17g, max |dx| %.3e" % (
                    r, s.num_iterations, s.final_cost, float(np.max(np.abs(x - ref[1])))))
    print("deterministic" if bad == 0 else "NON-DETERMINISTIC: %d of %d repeats differ" % (bad, args.repeats - 1))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
