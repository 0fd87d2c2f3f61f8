import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import helpers
from calico_amd import synthetic as syn
hip = helpers.hip_api()
sc = syn.config_scene(int(os.environ.get("CFG", "3")))
built = syn.build_problem(hip, sc)
P = built.problem
init = [(int(b), sc.ctrl[i].copy()) for i, b in enumerate(built.ctrl_blocks)]
for s, sb in zip(sc.sensors, built.sensor_blocks):
    init += [(sb["intrinsics"], s.intrinsics.copy()), (sb["t"], s.t.copy()), (sb["q"], s.q.copy()), (sb["latency"], np.array([s.latency]))]
ids = np.array([b for b, _ in init], np.int32); vals = np.concatenate([np.asarray(v, float).ravel() for _, v in init])
o = hip.default_options(); o.minimizer_progress_to_stdout = 0; o.sync_every = 8
def run(n):
    t0 = time.perf_counter(); P.set_param_blocks(ids, vals); t1 = time.perf_counter()
    o.max_num_iterations = n; s = P.solve(o); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, s.num_iterations, s.solve_time_in_seconds, s.total_time_in_seconds
run(50)
for n in (1, 2, 9, 17, 50, 50, 50):
    r = run(n)
    print("max_iter %2d: reset %.1f us  solve %.1f us  (%d iterations; lib solve_time %.1f us, total %.1f us) -> %.1f us/iter" % (n, r[0]*1e6, r[1]*1e6, r[2], r[3]*1e6, r[4]*1e6, r[1]*1e6/max(1,r[2])))
