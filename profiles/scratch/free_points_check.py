"""Free model points at configs[1] scale: HIP vs oracle (3 iterations), then HIP timing."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import helpers
from calico_amd import synthetic as syn, _capi

hip, ora = helpers.hip_api(), helpers.oracle_api()
sc = syn.make_scene(1, 1, False, cam_rate=20.0, duration=6.95, chart="april", seed=3, pixel_noise=0.1,
                    segment_duration=6.95 / 23.9, free_points=True)
print("blocks", sc.num_blocks, "points", len(sc.points))
g = syn.build_problem(hip, sc)
r = syn.build_problem(ora, sc)
for P, api in ((g, hip), (r, ora)):
    o = api.default_options(); o.max_num_iterations = 3; o.num_threads = 32; o.minimizer_progress_to_stdout = 0
    t = time.time(); s = P.problem.solve(o); dt = time.time() - t
    print(api is hip and "hip" or "oracle", s.initial_cost, s.final_cost, s.num_effective_parameters_reduced, "%.3fs" % dt)
ig, ir = g.problem.iterations(), r.problem.iterations()
for a, b in zip(ig, ir):
    print(a.iteration, a.cost, b.cost, abs(a.cost - b.cost) / abs(b.cost), a.step_is_successful, b.step_is_successful)
g2 = syn.build_problem(hip, sc)
o = hip.default_options(); o.max_num_iterations = 30; o.minimizer_progress_to_stdout = 0
o.function_tolerance = 0; o.gradient_tolerance = 0; o.parameter_tolerance = 0
g2.problem.solve(o)
g2 = syn.build_problem(hip, sc)
o1 = hip.default_options(); o1.max_num_iterations = 1; o1.minimizer_progress_to_stdout = 0
t = time.time(); g2.problem.solve(o1); print("first solve incl. finalize: %.3f s" % (time.time() - t))
t = time.time(); s = g2.problem.solve(o); dt = time.time() - t
print("hip 30 iterations: %.3f s -> %.1f it/s" % (dt, len(g2.problem.iterations()) / dt), s.final_cost)
