"""stochqn_amd -- MI355X-native stochastic quasi-Newton step (oLBFGS / SQN / adaQN).

The product is `lib/libstochqn.so`: hand-written HIP kernels (gfx950) plus the reverse-communication
state machines behind the reference's free-mode C ABI (include/stochqn.h).  This Python package is
only the host-side mirror of the reference's Python free-mode objects over that ABI (`free.py`) and
the loader below.  There is no CPU fallback: if the HIP library is missing, loading fails.
"""
import ctypes as _C
import os as _os

from . import _abi

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.path.join(_HERE, "lib", "libstochqn.so")

_bound = None
_cdll = None


def cdll():
    """The raw ctypes handle of libstochqn.so (raises if it has not been built)."""
    global _cdll
    if _cdll is None:
        if not _os.path.exists(LIB_PATH):
            raise ImportError(
                "stochqn_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C stochqn_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        _cdll = _C.CDLL(LIB_PATH, mode=_C.RTLD_GLOBAL)
    return _cdll


def lib():
    """The public C ABI (include/stochqn.h) bound with ctypes prototypes."""
    global _bound
    if _bound is None:
        _bound = _abi.Bound(cdll(), prefix="")
    return _bound


from .free import oLBFGS_free, SQN_free, adaQN_free  # noqa: E402

__all__ = ["lib", "cdll", "LIB_PATH", "oLBFGS_free", "SQN_free", "adaQN_free"]
