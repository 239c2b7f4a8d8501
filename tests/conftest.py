import ctypes as C
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
    # The parity tests compare EVERY array a call touches with the oracle's, the search direction left in `grad` included:
    # host callers get it copied back here (the library's default leaves a host caller's `grad` alone -- the reference
    # documents it as an input that is clobbered, and the copy is n words over PCIe; test_strict_grad_option_host_caller
    # covers the default).
    for use_float in (False, True):
        h = stochqn_amd.cdll(use_float)
        h.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
        assert h.stochqn_hip_set_option(b"strict_grad", 1.0) == 0
    return be
