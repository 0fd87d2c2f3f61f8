# Wall time of whole solves (parameter reset + calico_solve, as bench.py's timed region does) against their iteration count:
# slope = one LM iteration in steady state, intercept = what a solve costs beside its iterations. usage: boundary_fit.py [config]
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch; torch.zeros(1, device="cuda")
from calico_amd import _capi, synthetic as syn
api = _capi.load_hip()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scene = syn.config_scene(cfg)
b = syn.build_problem(api, scene)
init = [(int(bb), scene.ctrl[i].copy()) for i, bb in enumerate(b.ctrl_blocks)]
for s_, sb in zip(scene.sensors, b.sensor_blocks):
    init += [(sb["intrinsics"], s_.intrinsics.copy()), (sb["t"], s_.t.copy()), (sb["q"], s_.q.copy()), (sb["latency"], np.array([s_.latency]))]
ids = np.array([i for i, _ in init], np.int32); vals = np.concatenate([np.asarray(v, float).ravel() for _, v in init])
o = api.default_options(); o.minimizer_progress_to_stdout = 0
o.function_tolerance = 0.0; o.parameter_tolerance = 0.0; o.gradient_tolerance = 0.0
def run(n, reps=40):
    o.max_num_iterations = n
    ts = []
    for r in range(reps + 5):
        t = time.perf_counter(); b.problem.set_param_blocks(ids, vals); s = b.problem.solve(o); dt = time.perf_counter() - t
        if r >= 5: ts.append(dt * 1e6)
    ts.sort()
    return ts[len(ts) // 2], s.num_iterations
ns, ts = [], []
for n in (2, 5, 10, 15, 20, 25):
    t, it = run(n); ns.append(it); ts.append(t)
    print("iterations %3d  median %.1f us  (%.2f us per iteration)" % (it, t, t / it), flush=True)
A = np.vstack([np.array(ns, float), np.ones(len(ns))]).T
slope, icpt = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
print("fit: %.2f us per iteration + %.1f us per solve" % (slope, icpt))
