"""Free-mode optimiser objects over the C ABI -- the host-side mirror of the reference's Python layer.

Mirrors reference stochqn/_optimizers.py:791-1364: the `_*_holder` classes that own every state
array (sizes as in `_optimizers.py:791-879`, including the 1-element place-holders), and the
`oLBFGS_free / SQN_free / adaQN_free` objects with `run_optimizer`, `update_gradient`,
`update_hess_vec`, `update_function` and the same request dictionary.  Per call the C structs are
rebuilt from the Python-side arrays and counters, and the counters are copied back afterwards,
exactly as reference stochqn/pywrapper.pxi:89-207 does ("profile B" of SURVEY.md section 8b).

Two memory spaces are supported:
  * ``space="host"``   -- numpy arrays, what the reference's Python package passes;
  * ``space="device"`` -- torch double tensors on the GPU; the library then works on them in place.

The arithmetic is done by whatever library `backend` wraps: `stochqn_amd.lib()` (the HIP library;
default) or, in the test-suite only, the CPU oracle.
"""
import ctypes as C
import mmap
import weakref

import numpy as np

from . import _abi


# ----------------------------------------------------------------------------------------------
# array spaces
# ----------------------------------------------------------------------------------------------
def _unpin(lib, ptr):
    try:
        lib.stochqn_hip_unpin_host(C.c_void_p(ptr))
    except Exception:                                   # interpreter shutdown: the process is going away anyway
        pass


class _HostSpace:
    """numpy arrays.  Arrays that cross PCIe on every call -- the object's own gradient / hess_vec / x_sum / x_avg_prev and
    the user's x -- are page-locked through stochqn_hip_pin_host for exactly as long as the array object lives: a
    `weakref.finalize` on the ndarray unpins the range when the array is collected, BEFORE numpy frees its memory
    (weak-reference callbacks run at the start of deallocation).  That is the guarantee the library cannot give itself
    behind the C ABI (stochqn_hip.h: "host arrays pinned by their owner"), and what makes pinning safe here.

    The arrays the object makes for itself (`empty`) get an anonymous mapping of their OWN once they are large enough to be
    pinned: they start on a page boundary, share no page with anything else and are unmapped as a whole when the array dies
    (after its finaliser has unpinned it) -- whatever the process's malloc thresholds are.  numpy's own allocations of that
    size may sit in the program-break heap (glibc raises its mmap threshold up to 32 MiB as large blocks are freed), where a
    block shares its first and last page with its neighbours and the break moves under it; the library declines to pin such a
    range (stochqn_hip_pin_host returns 1) and the user's x, if it lives there, crosses the link through the runtime's
    pageable path instead."""
    name = "host"
    PIN_MIN_BYTES = 4 << 20                             # below, the runtime's staged copies are as fast

    def __init__(self, use_float=False):
        self.dtype = np.float32 if use_float else np.float64
        self._lib = None
        self._pins = {}                                 # address -> (weak reference to the array, its finaliser)
        self._left = {}                                 # address -> (weak reference, bytes) of arrays the library left pageable: not asked again

    def attach(self, backend):
        lib = getattr(backend, "lib", None)
        if lib is not None and not getattr(backend, "prefix", "") and hasattr(lib, "stochqn_hip_pin_host"):
            lib.stochqn_hip_pin_host.argtypes = [C.c_void_p, C.c_size_t]
            lib.stochqn_hip_unpin_host.argtypes = [C.c_void_p]
            self._lib = lib

    def pin(self, a):
        """Page-lock `a` for its lifetime (no-op for small, non-contiguous or foreign arrays, and without the HIP library)."""
        if self._lib is None or not isinstance(a, np.ndarray) or a.nbytes < self.PIN_MIN_BYTES or not a.flags.c_contiguous:
            return
        ptr = a.ctypes.data
        for p, (ref, fin) in list(self._pins.items()):  # an array that was resized in place sits elsewhere now: let the old range go
            obj = ref()
            if obj is None or obj.ctypes.data != p:
                fin()
                del self._pins[p]
        if ptr in self._pins:
            return
        left = self._left.get(ptr)
        if left is not None and left[0]() is a and left[1] == a.nbytes:
            return                                      # the same array, declined before (break heap, pages shared, pinned by its owner)
        if (ptr & 4095) > 64:
            # an array that begins deep inside a page shares that page with whatever precedes it: a block of some heap (the library
            # recognises glibc's heaps by itself -- the program break, the thread arenas -- but no other allocator's pools).  Arrays
            # with a mapping of their own begin within a malloc header of a page boundary.  Left pageable.
            self._leave(ptr, a)
            return
        if self._lib.stochqn_hip_pin_host(C.c_void_p(ptr), C.c_size_t(a.nbytes)) == 0:
            self._pins[ptr] = (weakref.ref(a), weakref.finalize(a, _unpin, self._lib, ptr))
        else:
            self._leave(ptr, a)

    def _leave(self, ptr, a):
        if len(self._left) >= 32:                       # a caller that brings a new x every call: the dead ones go
            self._left = {p: v for p, v in self._left.items() if v[0]() is not None}
        self._left[ptr] = (weakref.ref(a), a.nbytes)

    def unpin_all(self):
        for _, fin in self._pins.values():
            fin()
        self._pins = {}

    def empty(self, n):
        # the reference uses np.empty; zeros keeps runs reproducible and is a legal instance of it
        n = int(n)
        nbytes = n * np.dtype(self.dtype).itemsize
        if self._lib is not None and nbytes >= self.PIN_MIN_BYTES:
            return np.frombuffer(mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE),
                                 dtype=self.dtype, count=n)      # a PRIVATE anonymous mapping of its own, zero-filled by the kernel (the default of mmap.mmap(-1, ...) is MAP_SHARED: a fork()ed child would share and mutate the parent's optimiser state)
        return np.zeros(n, dtype=self.dtype)

    zeros = empty

    @staticmethod
    def ptr(a):
        return a.ctypes.data

    def assign(self, dst, src):
        dst[:] = np.asarray(src, dtype=self.dtype).reshape(-1)

    def is_array(self, a):
        return isinstance(a, np.ndarray) and a.dtype == self.dtype


class _DeviceSpace:
    name = "device"

    def attach(self, backend):
        pass

    def pin(self, a):
        pass

    def unpin_all(self):
        pass

    def __init__(self, device=None, use_float=False):
        import torch
        self.torch = torch
        self.device = torch.device(device if device is not None else "cuda")
        self.dtype = np.float32 if use_float else np.float64
        self.tdtype = torch.float32 if use_float else torch.float64

    def empty(self, n):
        return self.torch.zeros(int(n), dtype=self.tdtype, device=self.device)

    zeros = empty

    @staticmethod
    def ptr(a):
        return a.data_ptr()

    def assign(self, dst, src):
        if not isinstance(src, self.torch.Tensor):
            src = self.torch.as_tensor(np.asarray(src, dtype=self.dtype))
        dst.copy_(src.reshape(-1))

    def is_array(self, a):
        return isinstance(a, self.torch.Tensor) and a.dtype == self.tdtype and a.is_cuda


def _space(space, device=None, use_float=False):
    if space == "host":
        return _HostSpace(use_float)
    if space == "device":
        return _DeviceSpace(device, use_float)
    raise ValueError("space must be 'host' or 'device'")


# ----------------------------------------------------------------------------------------------
# state holders (reference stochqn/_optimizers.py:791-879)
# ----------------------------------------------------------------------------------------------
class _BFGS_mem_holder:
    def __init__(self, sp, mem_size, n, min_curvature, y_reg, upd_freq):
        self.s_mem = sp.empty(n * mem_size)
        self.y_mem = sp.empty(n * mem_size)
        self.buffer_rho = np.zeros(mem_size, dtype=sp.dtype)      # tiny, always host: read back by callers
        self.buffer_alpha = np.zeros(mem_size, dtype=sp.dtype)
        k = n if min_curvature > 0 else 1
        self.s_bak = sp.empty(k)
        self.y_bak = sp.empty(k)
        self.mem_size, self.mem_used, self.mem_st_ix = int(mem_size), 0, 0
        self.upd_freq = int(upd_freq)
        self.y_reg, self.min_curvature = float(y_reg), float(min_curvature)

    def c_struct(self, sp, be):
        return be.bfgs_mem(sp.ptr(self.s_mem), sp.ptr(self.y_mem),
                             self.buffer_rho.ctypes.data, self.buffer_alpha.ctypes.data,
                             sp.ptr(self.s_bak), sp.ptr(self.y_bak),
                             self.mem_size, self.mem_used, self.mem_st_ix, self.upd_freq,
                             self.y_reg, self.min_curvature)


class _Fisher_mem_holder:
    def __init__(self, sp, mem_size, n):
        self.F = sp.empty(n * mem_size)
        self.buffer_y = np.zeros(mem_size, dtype=sp.dtype)
        self.mem_size, self.mem_used, self.mem_st_ix = int(mem_size), 0, 0

    def c_struct(self, sp, be):
        return be.fisher_mem(sp.ptr(self.F), self.buffer_y.ctypes.data,
                               self.mem_size, self.mem_used, self.mem_st_ix)


_TASK = _abi.TASKS
_INFO = _abi.INFOS


class _StochQN_free:
    """Shared argument handling (reference stochqn/_optimizers.py:881-935)."""

    def _take_common_inputs(self, mem_size, min_curvature, y_reg, check_nan, nthreads, space, device, backend,
                            use_float=False):
        assert isinstance(mem_size, int) and mem_size > 0
        if min_curvature is not None:
            assert min_curvature > 0
        else:
            min_curvature = 0
        if y_reg is not None:
            assert y_reg > 0
        else:
            y_reg = 0
        if nthreads is None or nthreads <= 0:
            nthreads = 1
        self.mem_size = mem_size
        self.min_curvature = min_curvature
        self.y_reg = y_reg
        self.check_nan = bool(check_nan)
        self.nthreads = int(nthreads)
        self.use_float = bool(use_float)
        self._sp = _space(space, device, self.use_float)
        if backend is None:
            from . import lib
            backend = lib(use_float=self.use_float)
        assert bool(getattr(backend, "use_float", False)) == self.use_float, "backend precision mismatch"
        self._be = backend
        self._sp.attach(backend)
        self.initialized = False

    def update_gradient(self, gradient):
        """Hand over the gradient that the last request asked for."""
        self._sp.assign(self.gradient, gradient)

    # -- state that lives behind the ABI ---------------------------------------------------------
    def export(self):
        """Bring this object's numpy arrays up to date with the device (host space only).

        The reference's Python objects ARE their state (every array is a numpy array the C code works on
        in place), so pickling one is a checkpoint.  Here S, Y, F, x_sum, grad_sum_sq ... of a host-space
        object live in HBM between calls and the numpy arrays go stale; `export` copies them back
        (stochqn_hip_export).  Called automatically by pickle / copy (`__getstate__`)."""
        if getattr(self, "initialized", False) and self._sp.name == "host" and hasattr(self._be.lib, "stochqn_hip_export"):
            rc = self._be.lib.stochqn_hip_export(C.c_void_p(self._sp.ptr(self.BFGS_mem.s_mem)))
            if rc not in (0, -1000):                 # -1000: no device state exists (nothing ran yet): arrays are current
                raise RuntimeError("stochqn_hip_export failed")

    def __getstate__(self):
        if self._sp.name != "host":
            raise TypeError("only host-space optimizer objects can be pickled (device tensors belong to the caller)")
        if getattr(self._be, "prefix", ""):
            raise TypeError("only objects driven by libstochqn itself can be pickled")
        self.export()
        state = dict(self.__dict__)
        state["_be"] = None                          # ctypes handles do not pickle: re-bound on load
        state["_sp"] = None
        return state

    def __setstate__(self, state):
        from . import lib
        self.__dict__.update(state)
        self._sp = _space("host", None, self.use_float)
        self._be = lib(use_float=self.use_float)
        self._sp.attach(self._be)
        self._pin_own()

    def _order_streams(self):
        """Device space: the library's stream is ordered after the NULL stream only (stochqn_hip.h).  A caller
        that produced x / the gradient on another torch stream gets that stream drained first."""
        if self._sp.name == "device":
            torch = self._sp.torch
            cur = torch.cuda.current_stream(self._sp.device)
            if cur != torch.cuda.default_stream(self._sp.device):
                cur.synchronize()

    def release(self):
        """Free the device context that mirrors this object's arrays (the reference's Python
        objects have no such call; the library also recognises a recycled address by itself)."""
        if getattr(self, "initialized", False) and hasattr(self._be.lib, "stochqn_hip_release"):
            import ctypes
            self._be.lib.stochqn_hip_release(ctypes.c_void_p(self._sp.ptr(self.BFGS_mem.s_mem)))
        if getattr(self, "_sp", None) is not None:
            self._sp.unpin_all()

    def _pin_own(self):
        """The arrays of this object that travel on every call (host space): page-locked while they live."""
        for name in ("gradient", "hess_vec", "x_sum", "x_avg_prev"):
            a = getattr(self, name, None)
            if a is not None:
                self._sp.pin(a)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------------------------
    def _check_x(self, x):
        if not self._sp.is_array(x):
            raise ValueError("'x' has wrong dtype or lives in the wrong memory space.")
        if not self.initialized:
            self._initialize(int(x.shape[0]))
        self._pin_own()
        self._sp.pin(x)

    def _resolve(self, ptr, x, candidates):
        """Map a returned `*req` address back onto the array it aliases."""
        n = self._n
        if ptr == self._sp.ptr(x):
            return x
        for arr in candidates:
            base = self._sp.ptr(arr)
            size = arr.shape[0]
            esz = 4 if self.use_float else 8
            if base <= ptr < base + esz * size:
                off = (ptr - base) // esz
                return arr[off:off + n]
        raise RuntimeError("library returned a request pointer outside the caller's arrays")

    @staticmethod
    def _request(task, req, x_changed, niter, info):
        return {"task": _TASK[task], "requested_on": req,
                "info": {"x_changed_in_run": bool(x_changed), "iteration_number": int(niter),
                         "iteration_info": _INFO[info]}}


class oLBFGS_free(_StochQN_free):
    """oLBFGS optimizer, free mode (reference stochqn/_optimizers.py:937-1044)."""

    def __init__(self, mem_size=10, hess_init=None, min_curvature=1e-4, y_reg=None, check_nan=True,
                 nthreads=-1, use_float=False, space="host", device=None, backend=None):
        self._take_common_inputs(mem_size, min_curvature, y_reg, check_nan, nthreads, space, device, backend, use_float)
        if hess_init is not None:
            assert hess_init > 0
        else:
            hess_init = 0
        self.hess_init = hess_init

    def _initialize(self, n):
        sp = self._sp
        self._n = n
        self.BFGS_mem = _BFGS_mem_holder(sp, self.mem_size, n, self.min_curvature, self.y_reg, 1)
        self.grad_prev = sp.empty(n)
        self.niter, self.section = 0, 0
        self.gradient = sp.empty(n)
        self.initialized = True

    def run_optimizer(self, x, step_size):
        self._check_x(x)
        self._order_streams()
        sp = self._sp
        b = self.BFGS_mem.c_struct(sp, self._be)
        w = self._be.workspace_oLBFGS(C.pointer(b), sp.ptr(self.grad_prev), self.hess_init, self.niter,
                                  self.section, self.nthreads, int(self.check_nan), self._n)
        req, task, info = C.c_void_p(), C.c_int(), C.c_int()
        changed = self._be.run_oLBFGS(step_size, sp.ptr(x), sp.ptr(self.gradient), C.byref(req), C.byref(task),
                                      C.byref(w), C.byref(info))
        self.niter, self.section = w.niter, w.section
        self.BFGS_mem.mem_used, self.BFGS_mem.mem_st_ix = b.mem_used, b.mem_st_ix
        if changed == _abi.RECEIVED_INVALID_INPUT:
            raise ValueError("oLBFGS got an invalid workspace as input.")
        return self._request(task.value, self._resolve(req.value, x, []), changed, self.niter, info.value)


class SQN_free(_StochQN_free):
    """SQN optimizer, free mode (reference stochqn/_optimizers.py:1046-1190)."""

    def __init__(self, mem_size=10, bfgs_upd_freq=20, min_curvature=1e-4, y_reg=None, use_grad_diff=False,
                 check_nan=True, nthreads=-1, use_float=False, space="host", device=None, backend=None):
        self._take_common_inputs(mem_size, min_curvature, y_reg, check_nan, nthreads, space, device, backend, use_float)
        assert bfgs_upd_freq > 0
        self.bfgs_upd_freq = int(bfgs_upd_freq)
        self.use_grad_diff = bool(use_grad_diff)

    def _initialize(self, n):
        sp = self._sp
        self._n = n
        self.BFGS_mem = _BFGS_mem_holder(sp, self.mem_size, n, self.min_curvature, self.y_reg, self.bfgs_upd_freq)
        self.grad_prev = sp.empty(n if self.use_grad_diff else 1)
        self.x_sum = sp.zeros(n)
        self.x_avg_prev = sp.empty(n)
        self.niter, self.section = 0, 0
        self.gradient = sp.empty(n)
        self.hess_vec = sp.empty(1 if self.use_grad_diff else n)
        self.initialized = True

    def update_hess_vec(self, hess_vec):
        """Hand over the Hessian-vector product that the last request asked for."""
        self._sp.assign(self.hess_vec, hess_vec)

    def run_optimizer(self, x, step_size):
        self._check_x(x)
        self._order_streams()
        sp = self._sp
        b = self.BFGS_mem.c_struct(sp, self._be)
        w = self._be.workspace_SQN(C.pointer(b), sp.ptr(self.grad_prev), sp.ptr(self.x_sum), sp.ptr(self.x_avg_prev),
                               int(self.use_grad_diff), self.niter, self.section, self.nthreads,
                               int(self.check_nan), self._n)
        req, req_vec, task, info = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        changed = self._be.run_SQN(step_size, sp.ptr(x), sp.ptr(self.gradient), sp.ptr(self.hess_vec),
                                   C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
        self.niter, self.section = w.niter, w.section
        self.BFGS_mem.mem_used, self.BFGS_mem.mem_st_ix = b.mem_used, b.mem_st_ix
        if changed == _abi.RECEIVED_INVALID_INPUT:
            raise ValueError("SQN got an invalid workspace as input.")
        cands = [self.x_sum, self.x_avg_prev]
        r = self._resolve(req.value, x, cands)
        if _TASK[task.value] == "calc_hess_vec":
            r = (r, self._resolve(req_vec.value, x, [self.BFGS_mem.s_mem]))
        return self._request(task.value, r, changed, self.niter, info.value)


class adaQN_free(_StochQN_free):
    """adaQN optimizer, free mode (reference stochqn/_optimizers.py:1192-1364)."""

    def __init__(self, mem_size=10, fisher_size=100, bfgs_upd_freq=20, max_incr=1.01, min_curvature=1e-4,
                 scal_reg=1e-4, rmsprop_weight=None, y_reg=None, use_grad_diff=False, check_nan=True,
                 nthreads=-1, use_float=False, space="host", device=None, backend=None):
        self._take_common_inputs(mem_size, min_curvature, y_reg, check_nan, nthreads, space, device, backend, use_float)
        assert bfgs_upd_freq > 0
        if not use_grad_diff:
            assert fisher_size > 0
            fisher_size = int(fisher_size)
        else:
            fisher_size = 0
        if max_incr is not None:
            assert max_incr > 0
        else:
            max_incr = 0
        assert scal_reg > 0
        if rmsprop_weight is not None:
            assert 0 < rmsprop_weight < 1
        else:
            rmsprop_weight = 0
        self.fisher_size = fisher_size
        self.bfgs_upd_freq = int(bfgs_upd_freq)
        self.max_incr = max_incr
        self.scal_reg = scal_reg
        self.rmsprop_weight = rmsprop_weight
        self.use_grad_diff = bool(use_grad_diff)

    def _initialize(self, n):
        sp = self._sp
        self._n = n
        self.BFGS_mem = _BFGS_mem_holder(sp, self.mem_size, n, self.min_curvature, self.y_reg, self.bfgs_upd_freq)
        if self.use_grad_diff:
            self.Fisher_mem = _Fisher_mem_holder(sp, 1, 1)
            self.grad_prev = sp.empty(n)
        else:
            self.Fisher_mem = _Fisher_mem_holder(sp, self.fisher_size, n)
            self.grad_prev = sp.empty(1)
        self.H0 = sp.empty(n)
        self.x_sum = sp.zeros(n)
        self.x_avg_prev = sp.empty(n)
        self.grad_sum_sq = sp.zeros(n)
        self.f_prev = 0.0
        self.niter, self.section = 0, 0
        self.gradient = sp.empty(n)
        self.f = 0.0
        self.initialized = True

    def update_function(self, fun):
        """Hand over the objective value that the last request asked for."""
        self.f = float(fun)

    def run_optimizer(self, x, step_size):
        self._check_x(x)
        self._order_streams()
        sp = self._sp
        b = self.BFGS_mem.c_struct(sp, self._be)
        fm = self.Fisher_mem.c_struct(sp, self._be)
        w = self._be.workspace_adaQN(C.pointer(b), C.pointer(fm), sp.ptr(self.H0), sp.ptr(self.grad_prev),
                                 sp.ptr(self.x_sum), sp.ptr(self.x_avg_prev), sp.ptr(self.grad_sum_sq),
                                 self.f_prev, self.max_incr, self.scal_reg, self.rmsprop_weight,
                                 int(self.use_grad_diff), self.niter, self.section, self.nthreads,
                                 int(self.check_nan), self._n)
        req, task, info = C.c_void_p(), C.c_int(), C.c_int()
        changed = self._be.run_adaQN(step_size, sp.ptr(x), self.f, sp.ptr(self.gradient), C.byref(req),
                                     C.byref(task), C.byref(w), C.byref(info))
        self.niter, self.section, self.f_prev = w.niter, w.section, w.f_prev
        self.BFGS_mem.mem_used, self.BFGS_mem.mem_st_ix = b.mem_used, b.mem_st_ix
        self.Fisher_mem.mem_used, self.Fisher_mem.mem_st_ix = fm.mem_used, fm.mem_st_ix
        if changed == _abi.RECEIVED_INVALID_INPUT:
            raise ValueError("adaQN got an invalid workspace as input.")
        r = self._resolve(req.value, x, [self.x_sum, self.x_avg_prev])
        return self._request(task.value, r, changed, self.niter, info.value)
