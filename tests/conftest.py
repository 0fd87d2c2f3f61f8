import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import helpers
    helpers.build_oracle()
    return helpers.oracle_api()


@pytest.fixture(scope="session")
def hip():
    import helpers
    # PyTorch's HIP runtime (and its own librccl) first, as in bench.py: the library then shares them whatever the order
    # of the tests in the session (some tests create torch streams, others an RCCL communicator)
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    return helpers.hip_api()
