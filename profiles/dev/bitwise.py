import sys, hashlib, numpy as np
sys.path.insert(0, '/root/repo')
from calico_amd import _capi, synthetic as syn
api = _capi.load_hip()
for cfg in (1, 2, 3):
    scene = syn.config_scene(cfg)
    built = syn.build_problem(api, scene, device=0)
    P = built.problem
    o = api.default_options(); o.minimizer_progress_to_stdout = 0
    s = P.solve(o)
    vals = []
    for b in built.ctrl_blocks[:50]:
        vals.append(np.asarray(P.get_param_block(int(b), 6)))
    h = hashlib.sha256(np.concatenate(vals).tobytes()).hexdigest()[:16]
    print(cfg, s.num_iterations, repr(s.final_cost), h)
