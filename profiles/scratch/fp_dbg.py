import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import helpers
from calico_amd import synthetic as syn
hip = helpers.hip_api()
sc = syn.make_scene(1, 1, False, cam_rate=20.0, duration=6.95, chart="april", seed=3, pixel_noise=0.1, segment_duration=6.95 / 23.9, free_points=True)
g = syn.build_problem(hip, sc)
o = hip.default_options(); o.max_num_iterations = 2; o.minimizer_progress_to_stdout = 0
g.problem.solve(o)
