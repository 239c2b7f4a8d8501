"""stochqn_amd -- MI355X-native stochastic quasi-Newton step (oLBFGS / SQN / adaQN).

The product is `lib/libstochqn.so`: hand-written HIP kernels (gfx950) plus the reverse-communication
state machines behind the reference's free-mode C ABI (include/stochqn.h).  This Python package is
only the host-side mirror of the reference's Python free-mode objects over that ABI (`free.py`) and
the loader below.  There is no CPU fallback: if the HIP library is missing, loading fails.
"""
import ctypes as _C
import importlib.util as _ilu
import os as _os
import sys as _sys

from . import _abi

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.path.join(_HERE, "lib", "libstochqn.so")
LIB_PATH_F32 = _os.path.join(_HERE, "lib", "libstochqn_f32.so")

_bound = {}
_cdll = {}


def _share_hip_runtime_with_torch():
    """One ROCm runtime per process.  torch's wheels bundle their own libamdhip64.so, libhsa-runtime64.so and
    librccl.so (same SONAMEs as /opt/rocm's).  Measured on the GPU box (profiles/src/probe_order.py):
      * libstochqn.so loaded first brings in /opt/rocm's HIP; a later `import torch` adds the bundled one --
        two runtimes, torch.cuda then reports no device and torch's device pointers mean nothing to the library;
      * any librccl.so loaded before torch (libstochqn dlopen()s RCCL when a communicator is asked for; even
        torch's own copy preloaded by path) ends the process with a double free at exit.
    With torch imported first the dynamic loader resolves libstochqn's NEEDED / dlopen names to the bundled
    copies by SONAME and there is exactly one of each.  So: if torch is installed, import it before the
    library is loaded.  Without torch nothing happens (C / R callers link /opt/rocm directly)."""
    if "torch" in _sys.modules:
        return
    try:
        if _ilu.find_spec("torch") is not None:
            import torch  # noqa: F401
    except (ImportError, ValueError):
        pass


def cdll(use_float=False):
    """The raw ctypes handle of libstochqn.so / libstochqn_f32.so (raises if it has not been built).
    The two precisions export the same symbol names, so each is loaded with local binding."""
    key = bool(use_float)
    if key not in _cdll:
        path = LIB_PATH_F32 if key else _os.environ.get("STOCHQN_LIB", LIB_PATH)    # STOCHQN_LIB: A/B of library builds
        if not _os.path.exists(path):
            raise ImportError(
                "stochqn_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C stochqn_amd/csrc`.  There is no CPU fallback." % path)
        _share_hip_runtime_with_torch()
        _cdll[key] = _C.CDLL(path, mode=_C.RTLD_LOCAL)
    return _cdll[key]


def lib(use_float=False):
    """The public C ABI (include/stochqn.h) bound with ctypes prototypes, for one precision."""
    key = bool(use_float)
    if key not in _bound:
        _bound[key] = _abi.Bound(cdll(key), prefix="", use_float=key)
    return _bound[key]


from .free import oLBFGS_free, SQN_free, adaQN_free  # noqa: E402

__all__ = ["lib", "cdll", "LIB_PATH", "oLBFGS_free", "SQN_free", "adaQN_free"]
