"""Test helpers: loads the CPU oracle (test infrastructure) and the HIP library."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from calico_amd import _capi  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libcalico_oracle.so")
_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def oracle_lib():
    if not os.path.exists(ORACLE_LIB):
        build_oracle()
    return C.CDLL(ORACLE_LIB)


def oracle_api():
    """The CPU oracle behind the same ABI shape (prefix oracle_). Checker only."""
    global _oracle
    if _oracle is None:
        _oracle = _capi.CApi(oracle_lib(), "oracle_", has_device=False)
    return _oracle


def hip_api():
    return _capi.load_hip()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
