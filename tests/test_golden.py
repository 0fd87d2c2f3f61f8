"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): committed inputs with
the oracle's outputs. CPU: the oracle still reproduces them (guards the checker itself). GPU: the HIP
path reproduces them through the C ABI without the oracle in the loop."""
import glob
import os

import numpy as np
import pytest

import helpers
from calico_amd import synthetic as syn
from golden_io import scene_from_dict

FIXTURES = sorted(f for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
                  if not os.path.basename(f).startswith("ceres_log"))     # the reference's iteration table: tests/test_ceres_log.py


def _check(api, path, rtol_eval, rtol_solve):
    d = dict(np.load(path))
    scene = scene_from_dict(d)
    built = syn.build_problem(api, scene)
    cost, g, H = built.problem.evaluate()
    assert abs(cost - d["cost"]) <= rtol_eval * abs(d["cost"])
    assert np.abs(g - d["gradient"]).max() <= rtol_eval * np.abs(d["gradient"]).max()
    sg = np.sqrt(np.diag(d["jtj"]))
    sg = np.where(sg > 0, sg, 1.0)
    assert (np.abs(H - d["jtj"]) / np.outer(sg, sg)).max() <= rtol_eval
    for i, s in enumerate(scene.sensors):
        r, v = built.problem.residuals(built.sensor_ids[i], s.n, s.dim)
        assert np.abs(r - d["res%d" % i]).max() <= rtol_eval * max(1.0, np.abs(d["res%d" % i]).max())
        assert np.array_equal(built.problem.inlier_mask(built.sensor_ids[i], s.n, 3.0), d["mask%d" % i])  # bit exact
    o = api.default_options()
    o.minimizer_progress_to_stdout = 0
    o.max_num_iterations = 100
    sm = built.problem.solve(o)
    assert sm.termination_type == int(d["termination_type"])
    assert abs(sm.final_cost - d["final_cost"]) <= 1e-8 * d["final_cost"]
    est, ctrl = syn.read_back(built, scene)
    for i, e in enumerate(est):
        for key, gk in (("intrinsics", "intr_final%d"), ("q", "q_final%d"), ("t", "t_final%d")):
            ref = d[gk % i]
            assert np.abs(e[key] - ref).max() <= rtol_solve * max(1e-3, np.abs(ref).max()), (key, i)
        assert abs(e["latency"] - d["lat_final%d" % i][0]) <= rtol_solve * max(1e-3, abs(d["lat_final%d" % i][0]))
        assert np.array_equal(built.problem.inlier_mask(built.sensor_ids[i], scene.sensors[i].n, 3.0), d["mask_final%d" % i])
    assert np.abs(ctrl - d["ctrl_final"]).max() <= rtol_solve * max(1.0, np.abs(d["ctrl_final"]).max())
    if "points_final" in d:   # free model points
        pts = np.stack([built.problem.get_param_block(int(b), 3) for b in built.point_blocks])
        assert np.abs(pts - d["points_final"]).max() <= rtol_solve * max(1.0, np.abs(d["points_final"]).max())


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(path, oracle):
    _check(oracle, path, 1e-12, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_reproduces_golden(path, hip):
    # parameter estimates within 1e-6 relative (north star), integer masks bit exact
    _check(hip, path, 1e-9, 1e-6)


def test_fixtures_present():
    assert len(FIXTURES) >= 5
