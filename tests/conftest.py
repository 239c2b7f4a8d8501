import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _heap_mode():
    """Round 4 calmed the test process's heap here (a fixed mmap threshold, no trimming) after two GPU memory-access faults at
    break-heap addresses; that mask is GONE (round 5): the test process has glibc's own dynamic thresholds, like an R or Python
    session, and the product no longer page-locks ranges in the break heap or ranges that share a page with another pin
    (stochqn_amd/csrc/runtime.cpp: pinnable_in_place; stochqn_amd/free.py gives its own arrays mappings of their own).
    What remains is a DIAGNOSTIC switch for tools/suite_soak.sh: STOCHQN_TEST_HEAP=brk provokes the allocator (blocks of up to
    32 MiB in the break heap, which is cut back at every free); unset, nothing is touched."""
    mode = os.environ.get("STOCHQN_TEST_HEAP", "")
    if mode != "brk":
        return
    try:
        libc = C.CDLL("libc.so.6")
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        libc.mallopt(M_MMAP_THRESHOLD, 32 << 20)             # (mallopt takes an int)
        libc.mallopt(M_TRIM_THRESHOLD, 0)
    except OSError:
        pass


_TRACE = None


def _pin_trace():
    """A line per page-locked range that comes and goes (stochqn_amd/free.py through stochqn_hip_pin_host) and per test (its id, the
    program break), flushed as written, in gpurun_out/pin_trace.log: should the GPU fault of DESIGN.md 7.1 come back, the log says
    whether its address lay in a range that was page-locked at the time, in one that had been, or in none.  GPU runs only."""
    global _TRACE
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        _TRACE = open(os.path.join(ROOT, "gpurun_out", "pin_trace.log"), "w", buffering=1)
    except OSError:
        return
    from stochqn_amd import free
    pin0, unpin0 = free._HostSpace.pin, free._unpin

    def pin(self, a):
        before = set(self._pins)
        pin0(self, a)
        for p in set(self._pins) - before:
            _TRACE.write("pin   %#x +%d\n" % (p, a.nbytes))

    def unpin(lib, ptr):
        _TRACE.write("unpin %#x\n" % ptr)
        unpin0(lib, ptr)

    free._HostSpace.pin, free._unpin = pin, unpin


@pytest.fixture(autouse=True)
def _trace_test(request):
    if _TRACE is not None:
        libc = C.CDLL(None)
        libc.sbrk.restype, libc.sbrk.argtypes = C.c_void_p, [C.c_long]
        _TRACE.write("test  %s brk %#x\n" % (request.node.nodeid, libc.sbrk(0) or 0))
    yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _heap_mode()
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        _pin_trace()


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
