import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import helpers
from calico_amd import synthetic as syn
hip = helpers.hip_api()
sc = syn.config_scene(int(os.environ.get("CFG", "3")))
g = syn.build_problem(hip, sc)
o = hip.default_options(); o.max_num_iterations = 2; o.minimizer_progress_to_stdout = 0
g.problem.solve(o)
