# the largest difference between the estimates of the library (CALICO_HIP_LIB with CALICO_DEV=1) and the oracle's on the unfused-route
# scenes of tests/test_gpu_parity.py, solved to convergence with the default options; in units of the test's bound (1e-6 x scale)
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch; torch.zeros(1, device="cuda")
import helpers, test_gpu_parity as t
from calico_amd import synthetic as syn
hip, oracle = helpers.hip_api(), helpers.oracle_api()
for name in sorted(t._ROUTE_SCENES):
    scene = t._route_scene(name, seed=19)
    opts = dict(function_tolerance=float(sys.argv[1]), parameter_tolerance=float(sys.argv[2])) if len(sys.argv) > 2 else {}
    gpu, ref, sg, sr = t.solve_both(scene, hip, oracle, max_iter=200, **opts)
    eg, cg = syn.read_back(gpu, scene); er, cr = syn.read_back(ref, scene)
    worst = 0.0
    for a, b in zip(eg, er):
        for key in ("intrinsics", "t", "q"):
            worst = max(worst, np.abs(a[key] - b[key]).max() / (1e-6 * max(1e-3, np.abs(b[key]).max())))
    print("%-24s iterations %d / %d   worst estimate difference %.3f of the bound   control points %.3g" % (name, sg.num_iterations, sr.num_iterations, worst, np.abs(cg - cr).max()))
