"""Loader of the CPU parity oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg -- never by
anything under stochqn_amd/ (the product path has no CPU fallback).
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from stochqn_amd import _abi  # noqa: E402  (declarations only; does not load the HIP library)

SO = os.path.join(_HERE, "liboracle.so")
SO_F32 = os.path.join(_HERE, "liboracle_f32.so")


def build(force=False, use_float=False):
    src = os.path.join(_HERE, "stochqn_oracle.c")
    so = SO_F32 if use_float else SO
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
    return so


_lib = None
_bound = None
_bound_f32 = None


def bound_f32():
    """The float build of the oracle (liboracle_f32.so) behind the float ABI."""
    global _bound_f32
    if _bound_f32 is None:
        _bound_f32 = _abi.Bound(C.CDLL(build(use_float=True)), prefix="oracle_", use_float=True)
    return _bound_f32


def cdll():
    global _lib
    if _lib is None:
        # ORACLE_SO: another build of the same source (tests/test_oracle_sanitized.py: -fsanitize=address,undefined)
        _lib = C.CDLL(os.environ.get("ORACLE_SO") or build())
        d, vp, i, sz = C.c_double, C.c_void_p, C.c_int, C.c_size_t
        _lib.oracle_set_threads.argtypes = [i]
        _lib.oracle_get_threads.restype = i
        _lib.oracle_two_loop.restype = None
        _lib.oracle_two_loop.argtypes = [vp, i, vp, d, vp, vp, sz, sz, sz, vp, vp]
        _lib.oracle_diag_rescale.restype = None
        _lib.oracle_diag_rescale.argtypes = [vp, vp, vp, i, d, d]
        _lib.oracle_fisher_product.restype = None
        _lib.oracle_fisher_product.argtypes = [vp, sz, i, vp, vp, vp]
        _lib.oracle_take_step.restype = None
        _lib.oracle_take_step.argtypes = [d, i, vp, vp, C.POINTER(_abi.bfgs_mem), d, vp, d, vp, d, i, C.POINTER(i)]
        if hasattr(_lib, "oracle_bind_threads"):
            _lib.oracle_bind_threads.restype = i
            _lib.oracle_unbind_threads.restype = None
        if hasattr(_lib, "oracle_first_touch"):
            _lib.oracle_first_touch.restype = None
            _lib.oracle_first_touch.argtypes = [vp, sz]
        _lib.oracle_set_threads(usable_cpus())
    return _lib


def bound():
    """The oracle's run_*/initialize_*/dealloc_* bound with the same prototypes as the product."""
    global _bound
    if _bound is None:
        _bound = _abi.Bound(cdll(), prefix="oracle_")
    return _bound


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  On the GPU
    boxes os.cpu_count() is 256 but the quota is 16 CPUs; 64 OpenMP threads there ran the two-loop 6x
    slower than 16 (throttling), which would understate the CPU baseline."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if quota != "max":
            cpus = min(cpus, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())               # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cpus = min(cpus, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return cpus


def set_threads(n):
    cdll().oracle_set_threads(int(n))


def two_loop(q, H0, h0, Y, S, m, used, st):
    """In-place two-loop on numpy arrays; returns (rho, alpha). Y, S are [m*n] row-major."""
    import numpy as np
    n = q.shape[0]
    rho = np.zeros(m)
    alpha = np.zeros(m)
    cdll().oracle_two_loop(q.ctypes.data, n, None if H0 is None else H0.ctypes.data, float(h0),
                           Y.ctypes.data, S.ctypes.data, m, used, st, rho.ctypes.data, alpha.ctypes.data)
    return rho, alpha


def fisher_product(F, fu, s):
    import numpy as np
    n = s.shape[0]
    t = np.zeros(fu)
    y = np.zeros(n)
    cdll().oracle_fisher_product(F.ctypes.data, fu, n, s.ctypes.data, t.ctypes.data, y.ctypes.data)
    return t, y


# ---- "the reference's kind of BLAS" for bench.py's cpu_baseline leg (stochqn_oracle.c: oracle_use_cblas) ----------------------
_blas_dll = None


def find_openblas():
    """An OpenBLAS on this box with a CBLAS interface: the one scipy bundles (32-bit integers, symbols scipy_cblas_*), the one
    numpy bundles (64-bit integers, scipy_cblas_*64_), or a system libopenblas (cblas_*).  -> dict(path, ilp64, fmt) or None."""
    import glob
    import site
    roots = []
    for get in (getattr(site, "getsitepackages", None), lambda: [site.getusersitepackages()]):
        try:
            roots += list(get()) if get else []
        except Exception:
            pass
    roots += [p for p in sys.path if p.endswith(("site-packages", "dist-packages"))]
    for sub, fmt, ilp64 in (("scipy.libs", "scipy_%s", 0), ("numpy.libs", "scipy_%s64_", 1)):
        for root in dict.fromkeys(roots):
            for f in sorted(glob.glob(os.path.join(root, sub, "libscipy_openblas*.so"))):
                return {"path": f, "ilp64": ilp64, "fmt": fmt}
    for name in ("libopenblas.so.0", "libopenblas.so"):
        try:
            C.CDLL(name)
            return {"path": name, "ilp64": 0, "fmt": "%s"}
        except OSError:
            continue
    return None


def use_cblas(info, threads=None):
    """Route the oracle's ddot / daxpy / dscal / dnrm2 and the two dgemv of the Fisher product through the CBLAS of `info`
    (find_openblas), or back to its own loops (info = None).  -> what was loaded: {library, config, threads, ilp64} or None."""
    global _blas_dll
    lib = cdll()
    lib.oracle_use_cblas.restype = C.c_int
    lib.oracle_use_cblas.argtypes = [C.c_void_p] * 5 + [C.c_int]
    if info is None:
        lib.oracle_use_cblas(None, None, None, None, None, 0)
        return None
    _blas_dll = dll = C.CDLL(info["path"])
    fmt = info["fmt"]
    addr = [C.cast(getattr(dll, fmt % ("cblas_" + n)), C.c_void_p).value for n in ("ddot", "daxpy", "dscal", "dnrm2", "dgemv")]
    out = {"library": info["path"], "ilp64": bool(info["ilp64"]), "config": None, "threads": None}
    try:
        cfg = getattr(dll, fmt % "openblas_get_config")
        cfg.restype = C.c_char_p
        out["config"] = cfg().decode()
    except AttributeError:
        pass
    try:
        if threads:
            getattr(dll, fmt % "openblas_set_num_threads")(int(threads))
        out["threads"] = int(getattr(dll, fmt % "openblas_get_num_threads")())
    except AttributeError:
        pass
    if lib.oracle_use_cblas(*addr, int(info["ilp64"])) != 1:
        return None
    return out
