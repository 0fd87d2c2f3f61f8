"""The C++ host side (include/calico/calico.hpp): the reference's BatchOptimizer / Sensor API shape
above the C ABI. tests/cpp/toy_stereo_imu.cpp restates the reference's own C++ tests against it."""
import os
import subprocess

import pytest

import helpers

SRC = os.path.join(helpers.ROOT, "tests", "cpp", "toy_stereo_imu.cpp")
EXE = os.path.join(helpers.ROOT, "tests", "cpp", "build", "toy_stereo_imu")


def _build():
    import __graft_entry__ as g
    g.build_hip()
    helpers.build_oracle()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(helpers.ROOT, "calico_amd")
    oradir = os.path.join(helpers.ROOT, "oracle")  # only --host-only uses it (spline fit without a GPU)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unknown-pragmas", "-DCALICO_TEST_HOOKS", "-I", os.path.join(helpers.ROOT, "include"),
                           SRC, "-o", EXE, "-L", libdir, "-lcalico_hip", "-Wl,-rpath," + libdir,
                           "-L", oradir, "-lcalico_oracle", "-Wl,-rpath," + oradir])


def test_facade_compiles_and_host_logic():
    """Container semantics, status codes, parameter bookkeeping, spline fit, Project: no GPU needed."""
    _build()
    out = subprocess.run([EXE, "--host-only"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")


@pytest.mark.gpu
def test_toy_stereo_camera_and_imu_calibration_cpp():
    """batch_optimizer_test.cpp:32-213 through BatchOptimizer::Optimize on the GPU."""
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
