import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle import oracle
    return oracle.bound()


@pytest.fixture(scope="session")
def hip_backend():
    """The product library.  Fails (never skips to a fallback) when it is missing or no GPU is visible."""
    import stochqn_amd
    be = stochqn_amd.lib()
    assert stochqn_amd.cdll().stochqn_hip_available() == 1, "libstochqn.so sees no HIP device"
    return be
