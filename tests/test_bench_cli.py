"""bench.py's launcher contract on a box without (enough) GPUs: `--gpus N` must never silently run on fewer devices."""
import os
import subprocess
import sys

import pytest

import helpers

BENCH = os.path.join(helpers.ROOT, "bench.py")


def _gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


@pytest.mark.skipif(_gpus() >= 2, reason="needs a box with fewer than two GPUs")
def test_gpus_2_without_two_devices_exits_nonzero():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "only %d GPU(s) visible" % _gpus() in r.stderr
    assert r.stdout.strip() == ""           # no JSON line that could be mistaken for a measurement


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not match" in r.stderr and r.stdout.strip() == ""


@pytest.mark.skipif(_gpus() >= 1, reason="checks the no-GPU behaviour")
def test_single_gpu_run_without_a_gpu_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "needs a GPU" in r.stderr and r.stdout.strip() == ""
