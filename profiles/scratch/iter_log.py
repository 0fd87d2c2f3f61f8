import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import helpers
from calico_amd import synthetic as syn
hip = helpers.hip_api()
sc = syn.config_scene(3)
built = syn.build_problem(hip, sc)
o = hip.default_options(); o.minimizer_progress_to_stdout = 1; o.max_num_iterations = 50
s = built.problem.solve(o)
print(s.num_iterations, s.num_successful_steps, s.num_unsuccessful_steps, s.termination_type, s.message, s.num_jacobian_evaluations, s.num_cost_evaluations)
