import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch; torch.zeros(1, device="cuda")
from calico_amd import _capi, synthetic as syn
api = _capi.load_hip()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scene = syn.config_scene(cfg)
b = syn.build_problem(api, scene)
o = api.default_options(); o.minimizer_progress_to_stdout = 0; o.max_num_iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s = b.problem.solve(o)
print("iterations", s.num_iterations, "cost", s.final_cost)
