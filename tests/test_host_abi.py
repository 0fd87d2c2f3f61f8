"""CPU-side checks of the drop-in boundary: libcalico_hip.so loads and exports every symbol that
include/calico_hip.h declares, fails loudly without a GPU (no CPU fallback), and the host-side
partition rule / option defaults behave. No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
from calico_amd import _capi


@pytest.fixture(scope="module")
def hiplib():
    import __graft_entry__ as g
    g.build_hip()
    return C.CDLL(_capi.hip_library_path())


def test_library_exports_every_declared_symbol(hiplib):
    header = open(os.path.join(helpers.ROOT, "include", "calico_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(calico_[a-z_]+)\s*\(", header)) - {"calico_allreduce_fn"})
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(hiplib, name), name
    assert sorted("calico_" + n for n in _capi.ABI_SYMBOLS) == declared


def test_default_solver_options_match_reference(hiplib):
    """batch_optimizer.cpp:10-17 over Ceres defaults."""
    api = _capi.CApi(hiplib, "calico_")
    o = api.default_options()
    assert (o.max_num_iterations, o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance) == (50, 1e-8, 1e-10, 1e-10)
    assert (o.initial_trust_region_radius, o.min_relative_decrease, o.min_lm_diagonal, o.max_lm_diagonal) == (1e4, 1e-3, 1e-6, 1e32)
    assert o.minimizer_progress_to_stdout == 1 and o.jacobi_scaling == 1 and o.max_num_consecutive_invalid_steps == 5
    oo = helpers.oracle_api().default_options()
    for name, _ in _capi.SolverOptions._fields_:
        assert getattr(o, name) == getattr(oo, name), name


@pytest.mark.skipif(helpers.has_gpu(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_a_fallback(hiplib):
    api = _capi.CApi(hiplib, "calico_")
    with pytest.raises(_capi.CalicoError) as e:
        _capi.Problem(api, 0)
    assert e.value.code == _capi.INTERNAL


def test_product_never_links_the_oracle(hiplib):
    import subprocess
    out = subprocess.run(["ldd", _capi.hip_library_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for f in os.listdir(os.path.join(helpers.ROOT, "calico_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(helpers.ROOT, "calico_amd", f)).read()
            assert "oracle_lib" not in src and "libcalico_oracle" not in src, f
    for f in os.listdir(os.path.join(helpers.ROOT, "calico_amd", "csrc")):
        if f.endswith((".cpp", ".hip", ".hpp")):
            src = open(os.path.join(helpers.ROOT, "calico_amd", "csrc", f)).read()
            assert "oracle/" not in src and "oracle_" not in src, f


def test_shard_windows_partition_rule():
    """calico_amd/csrc/shard.hpp through the oracle's multi-rank emulation: every residual block is
    owned by exactly one rank, windows are contiguous in time and balanced."""
    from calico_amd import synthetic as syn
    oracle = helpers.oracle_api()
    scene = syn.make_scene(2, 1, True, 2, cam_rate=10.0, imu_rate=50.0, duration=4.0, segment_duration=4.0 / 23.9, max_cam_obs=3000)
    full = syn.build_problem(oracle, scene)
    c_full, g_full, H_full = full.problem.evaluate()
    for world in (2, 3, 5):
        costs, gs, Hs = [], [], []
        for r in range(world):
            b = syn.build_problem(oracle, scene)
            assert oracle.lib.oracle_problem_set_shard(b.problem.h, r, world) == 0
            c, g, H = b.problem.evaluate()
            costs.append(c); gs.append(g); Hs.append(H)
        assert abs(sum(costs) - c_full) <= 1e-12 * c_full
        assert np.abs(sum(gs) - g_full).max() <= 1e-10 * np.abs(g_full).max()
        assert np.abs(sum(Hs) - H_full).max() <= 1e-10 * np.abs(H_full).max()
        share = np.array(costs) > 0
        assert share.all()


def test_missing_rccl_is_a_clean_error():
    """No librccl on the machine (CALICO_RCCL_LIB names the one library to load): calico_comm_get_unique_id returns
    kInternal instead of crashing while it formats dlerror() (round-3 advice: dlerror() clears its state)."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from calico_amd import _capi; "
            "lib = ctypes.CDLL(_capi.hip_library_path()); buf = (ctypes.c_uint8 * 128)(); "
            "lib.calico_comm_get_unique_id.restype = ctypes.c_int32; "
            "st = lib.calico_comm_get_unique_id(buf); st2 = lib.calico_comm_get_unique_id(buf); print('status', st, st2)"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, CALICO_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "status 13 13" in r.stdout, r.stdout


def test_rolling_chief_offset_table_matches_the_band_layout(hiplib):
    """bcr_kernels.hip, g_roll_tab (host-built, no device needed): where a lane's tile entries sit in the band's storage. The reduce
    buffer holds H(row, column) of the band at [(column's control point) * k + distance][column component][row component] (36 doubles
    per 6x6 block; solve_dev.hpp band_entry), superblock I = control points 5 I .. 5 I + 4 starting at I * 5 * k * 36. Restated here
    independently for every spline order 1..6 and lane: the spine's entries (tiles (0,0), (0,1), (1,1) of the 32x32 superblock in the
    elimination's register layout: lane (l16, lk), register r <-> row l16 (+16), column lk + 4 r (+16), taken from the lower triangle),
    the rows of B^T (next superblock's row 16 qt + l16 against this one's column 16 h + lk + 4 r) and of A^T (this superblock's row
    16 h + lk + 4 r against the LEFT one's column 16 qt + l16, counted from the left superblock's storage), with the masks of the
    entries that exist (inside the band: distance < k; inside the 30 real rows)."""
    fn = hiplib.calico_debug_roll_table
    fn.restype = C.c_int32
    fn.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]
    out = (C.c_uint32 * 48)()
    assert fn(0, 0, out) != 0 and fn(7, 0, out) != 0 and fn(6, 64, out) != 0      # argument checks

    def pos(hi, lo, k):      # H(hi, lo), lo <= hi, rows counted from the start of lo's superblock; None outside the band
        d = hi // 6 - lo // 6
        return 8 * (((lo // 6) * k + d) * 36 + (lo % 6) * 6 + hi % 6) if d < k else None

    for k in range(1, 7):
        for lane in range(64):
            assert fn(k, lane, out) == 0
            t = list(out)
            l16, lk = lane & 15, lane >> 4
            ok_s, ok_b, ok_a = t[28] & 0xffff, t[28] >> 16, t[29]
            for r in range(4):
                c = lk + 4 * r
                hi, lo = max(l16, c), min(l16, c)
                for e, (h, l) in ((r, (hi, lo)), (4 + r, (16 + l16, c)), (8 + r, (16 + hi, 16 + lo))):
                    want = pos(h, l, k)
                    exists = want is not None and h < 30
                    assert bool((ok_s >> e) & 1) == exists, (k, lane, e)
                    if exists:
                        assert t[e] == want, (k, lane, e)
                        assert t[e] < 8 * 5 * k * 36             # inside the superblock's own storage
            for qt in range(2):
                for h in range(2):
                    for r in range(4):
                        e = (qt * 2 + h) * 4 + r
                        cb, rn = 16 * h + lk + 4 * r, 16 * qt + l16      # B^T: this block's column, the next block's row
                        want = pos(30 + rn, cb, k)
                        exists = want is not None and rn < 30 and cb < 30
                        assert bool((ok_b >> e) & 1) == exists, (k, lane, "B", e)
                        if exists:
                            assert t[12 + e] == want
                        rb, cs = 16 * h + lk + 4 * r, 16 * qt + l16      # A^T: this block's row, the left block's column
                        want = pos(30 + rb, cs, k)
                        exists = want is not None and rb < 30 and cs < 30
                        assert bool((ok_a >> e) & 1) == exists, (k, lane, "A", e)
                        if exists:
                            assert t[32 + e] == want
    # spline order 6: five control points per superblock never leave the band inside a superblock; the coupling to the next one exists
    # exactly where the next block's control point is not behind this one's (distance 5 + j' - j < 6)
    assert fn(6, 0, out) == 0 and (out[28] & 0xfff) == 0xfff
