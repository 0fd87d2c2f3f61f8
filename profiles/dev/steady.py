# Marginal cost of one LM iteration inside a long solve (no solve boundary in the figure): two solves of N0 and N1 iterations from
# the same start with every tolerance at zero; (T1 - T0) / (N1 - N0). usage: steady.py [config] [N0] [N1] [rounds]
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch; torch.zeros(1, device="cuda")
from calico_amd import _capi, synthetic as syn
api = _capi.load_hip()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n1 = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
scene = syn.config_scene(cfg)
b = syn.build_problem(api, scene)
import numpy as np
init = [(int(bb), scene.ctrl[i].copy()) for i, bb in enumerate(b.ctrl_blocks)]
for s_, sb in zip(scene.sensors, b.sensor_blocks):
    init += [(sb["intrinsics"], s_.intrinsics.copy()), (sb["t"], s_.t.copy()), (sb["q"], s_.q.copy()), (sb["latency"], np.array([s_.latency]))]
ids = np.array([i for i, _ in init], np.int32); vals = np.concatenate([np.asarray(v, float).ravel() for _, v in init])
def run(n):
    o = api.default_options(); o.minimizer_progress_to_stdout = 0; o.max_num_iterations = n
    o.function_tolerance = 0.0; o.parameter_tolerance = 0.0; o.gradient_tolerance = 0.0
    b.problem.set_param_blocks(ids, vals)
    torch.cuda.synchronize()
    t = time.perf_counter(); s = b.problem.solve(o); torch.cuda.synchronize(); dt = time.perf_counter() - t
    return dt, s.num_iterations
run(n0); run(n1)
res = []
for r in range(rounds):
    t0, i0 = run(n0); t1, i1 = run(n1)
    res.append((t1 - t0) / max(1, i1 - i0) * 1e6)
    print("round", r, "N0", i0, "%.1f us" % (t0 * 1e6), "N1", i1, "%.1f us" % (t1 * 1e6), "marginal %.2f us / iteration" % res[-1], flush=True)
res.sort(); print("steady-state median %.2f us per iteration" % res[len(res) // 2])
