"""Shared drivers for the parity tests: scripted optimisation problems run call by call through a
free-mode object, recording everything observable through the C ABI after every call."""
import numpy as np

from stochqn_amd.free import oLBFGS_free, SQN_free, adaQN_free


_STAGE = {}
_STAGE_BYTES = 64 << 20


def _stage(torch, dtype):
    """A page-locked 64 MiB staging tensor per dtype.  Large test temporaries cross the link in chunks through it: DMA at
    link speed instead of the runtime's pageable path, which for big transfers page-locks the (about to be freed) temporary
    on the fly."""
    if dtype not in _STAGE:
        _STAGE[dtype] = torch.empty(_STAGE_BYTES // torch.empty(0, dtype=dtype).element_size(), dtype=dtype, pin_memory=True)
    return _STAGE[dtype]


def to_np(a):
    if isinstance(a, np.ndarray):
        return a.copy()
    a = a.detach()
    if not a.is_cuda:
        return a.numpy().copy()
    if a.numel() * a.element_size() <= _STAGE_BYTES or not a.is_contiguous():
        return a.cpu().numpy()                                     # .cpu() of a device tensor is a copy already
    import torch
    flat, st = a.reshape(-1), _stage(torch, a.dtype)
    out = np.empty(flat.numel(), dtype=st.numpy().dtype)
    for lo in range(0, flat.numel(), st.numel()):
        m = min(st.numel(), flat.numel() - lo)
        st[:m].copy_(flat[lo:lo + m])
        out[lo:lo + m] = st[:m].numpy()
    return out.reshape(tuple(a.shape))


def own_mapping(a):
    """A copy of `a` in an anonymous mapping of its own -- what malloc / numpy / R hand out for a large array when the allocator's
    mmap threshold is below its size, made deterministic: pages that belong to this array alone, unmapped when it dies.  The
    library page-locks such an array in place when asked to; one that sits in the program-break heap it leaves pageable
    (stochqn_amd/csrc/runtime.cpp: pinnable_in_place)."""
    import mmap
    a = np.ascontiguousarray(a)
    out = np.frombuffer(mmap.mmap(-1, max(a.nbytes, 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE), dtype=a.dtype, count=a.size).reshape(a.shape)
    out[...] = a
    return out


def to_dev(a, device="cuda"):
    """A numpy array as a new device tensor (large ones in chunks through the page-locked staging tensor)."""
    import torch
    a = np.ascontiguousarray(a)
    if a.nbytes <= _STAGE_BYTES:
        return torch.from_numpy(a).to(device)
    flat = torch.from_numpy(a.reshape(-1))
    st = _stage(torch, flat.dtype)
    out = torch.empty(flat.numel(), dtype=flat.dtype, device=device)
    for lo in range(0, flat.numel(), st.numel()):
        m = min(st.numel(), flat.numel() - lo)
        st[:m].copy_(flat[lo:lo + m])
        out[lo:lo + m].copy_(st[:m])                                # synchronous for the host: the stage is free again
    return out.reshape(tuple(a.shape))


# ------------------------------------------------------------------------------------------------
# problems
# ------------------------------------------------------------------------------------------------
class Rosenbrock2D:
    """fr / grr / Hvr of reference R/optimizers_free.R:56-66,181-195."""
    n = 2

    @staticmethod
    def x0():
        return np.array([0.0, 2.0])

    @staticmethod
    def f(x):
        return 100 * (x[1] - x[0] * x[0]) ** 2 + (1 - x[0]) ** 2

    @staticmethod
    def grad(x, call):
        return np.array([-400 * x[0] * (x[1] - x[0] * x[0]) - 2 * (1 - x[0]), 200 * (x[1] - x[0] * x[0])])

    @staticmethod
    def hess_vec(x, v):
        H = np.array([[1200 * x[0] ** 2 - 400 * x[1] + 2, -400 * x[0]], [-400 * x[0], 200.0]])
        return H @ v


class RosenbrockND:
    """rosen / rosen_der / rosen_hess_prod of reference example/c_rosen.c:13-60."""

    def __init__(self, n):
        self.n = n

    def f(self, x):
        return float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2))

    def grad(self, x, call=0):
        n = self.n
        out = np.zeros(n)
        out[0] = -400.0 * x[0] * (x[1] - x[0] * x[0]) - 2.0 * (1.0 - x[0])
        out[n - 1] = 200.0 * (x[n - 1] - x[n - 2] * x[n - 2])
        for i in range(1, n - 1):
            d1 = 200.0 * (x[i] - x[i - 1] * x[i - 1])
            d2 = 400.0 * (x[i + 1] - x[i] * x[i]) * x[i]
            d3 = 2.0 * (1.0 - x[i])
            out[i] = d1 - d2 - d3
        return out

    def hess_vec(self, x, p):
        n = self.n
        out = np.zeros(n)
        out[0] = (1200 * x[0] * x[0] - 400 * x[1] + 2.0) * p[0] - 400 * x[0] * p[0]
        out[n - 1] = -400.0 * x[n - 2] * p[n - 2] + 200.0 * p[n - 1]
        for i in range(1, n - 1):
            d1 = -400.0 * x[i - 1] * p[i - 1]
            d2 = (202 + 1200 * x[i] * x[i] - 400 * x[i + 1]) * p[i]
            d3 = 400.0 * x[i] * p[i + 1]
            out[i] = d1 + d2 - d3
        return out


class NoisyQuadratic:
    """f(x) = 1/2 sum d_i x_i^2 with multiplicative gradient noise that depends only on
    (seed, call index), so two backends see identical 'stochastic' gradients."""

    def __init__(self, n, seed=0, noise=0.01, nan_calls=(), f_spike_calls=()):
        rng = np.random.default_rng(seed)
        self.n = n
        self.d = 0.5 + rng.random(n)
        self._x0 = 1.0 + rng.random(n)
        self.seed = seed
        self.noise = noise
        self.nan_calls = set(nan_calls)
        self.f_spike_calls = set(f_spike_calls)
        self._factors = {}                              # call -> 1 + noise (2u - 1), kept for large n: tests drive the same calls two or three times

    def x0(self):
        return self._x0.copy()

    def f(self, x, call=0):
        v = 0.5 * float(np.sum(self.d * x * x))
        return v * 10.0 if call in self.f_spike_calls else v

    def grad(self, x, call):
        f = self._factors.get(call)
        if f is None:
            u = np.random.default_rng([self.seed, call]).random(self.n)
            f = 1.0 + self.noise * (2.0 * u - 1.0)
            if self.n >= 1_000_000 and 8 * self.n * (len(self._factors) + 1) <= (2 << 30):
                self._factors[call] = f
        g = self.d * x * f                              # (d x) f: the same operations in the same order with or without the cache
        if call in self.nan_calls:
            g = g.copy()
            g[self.n // 2] = np.nan
        return g

    def hess_vec(self, x, v):
        return self.d * v


# ------------------------------------------------------------------------------------------------
# trace driver
# ------------------------------------------------------------------------------------------------
def _req_id(opt, x, req):
    """Which caller array does the request alias?"""
    cands = {"x": x}
    for name in ("x_sum", "x_avg_prev"):
        if hasattr(opt, name):
            cands[name] = getattr(opt, name)
    sp = opt._sp
    p = sp.ptr(req)
    for name, arr in cands.items():
        if p == sp.ptr(arr):
            return name
    return "other"


def run_trace(opt, problem, x, step, ncalls, step_fn=None):
    """Drive `opt` for `ncalls` run_optimizer calls.  Returns a list of per-call records."""
    trace = []
    last_grad_x = 999983        # a resumed run may start with a same-batch request
    for call in range(ncalls):
        st = step_fn(call) if step_fn else step
        r = opt.run_optimizer(x, st)
        task = r["task"]
        req = r["requested_on"]
        rec = {
            "task": task,
            "info": r["info"]["iteration_info"],
            "changed": r["info"]["x_changed_in_run"],
            "niter": r["info"]["iteration_number"],
            "section": opt.section,
            "mem_used": opt.BFGS_mem.mem_used,
            "mem_st_ix": opt.BFGS_mem.mem_st_ix,
            "x": to_np(x),
        }
        if hasattr(opt, "Fisher_mem"):
            rec["f_used"] = opt.Fisher_mem.mem_used
            rec["f_st"] = opt.Fisher_mem.mem_st_ix
            rec["f_prev"] = opt.f_prev
        if task == "calc_hess_vec":
            rx, rv = req
            rec["req_id"] = _req_id(opt, x, rx)
            rec["req"] = to_np(rx)
            rec["req_vec"] = to_np(rv)
            opt.update_hess_vec(problem.hess_vec(to_np(rx), to_np(rv)))
        else:
            rec["req_id"] = _req_id(opt, x, req)
            rec["req"] = to_np(req)
            if task in ("calc_grad", "calc_grad_same_batch", "calc_grad_big_batch"):
                # same-batch gradients reuse the noise of the previous calc_grad call
                if task == "calc_grad":
                    last_grad_x = call
                noise_call = last_grad_x if task == "calc_grad_same_batch" else call
                opt.update_gradient(problem.grad(to_np(req), noise_call))
            elif task == "calc_fun_val_batch":
                opt.update_function(problem.f(to_np(req), call) if isinstance(problem, NoisyQuadratic)
                                    else problem.f(to_np(req)))
        trace.append(rec)
    return trace


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not np.all(np.isfinite(b)):
        return 0.0 if np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(
            a[np.isfinite(b)], b[np.isfinite(b)], rtol=1e-10, atol=0) else np.inf
    nb = np.linalg.norm(b)
    if nb == 0:
        return float(np.linalg.norm(a))
    return float(np.linalg.norm(a - b) / nb)


INT_KEYS = ("task", "info", "changed", "niter", "section", "mem_used", "mem_st_ix", "req_id", "f_used", "f_st")
VEC_KEYS = ("x", "req", "req_vec")


def compare_traces(got, want, tol=1e-10):
    """Integer-exact on everything discrete, norm-wise relative `tol` on every vector (fp64)."""
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        for k in INT_KEYS:
            if k in w:
                assert g[k] == w[k], "call %d: %s differs: %r vs %r" % (i, k, g[k], w[k])
        for k in VEC_KEYS:
            if k in w:
                e = rel_err(g[k], w[k])
                assert e <= tol, "call %d: %s rel err %.3e > %.1e" % (i, k, e, tol)
        if "f_prev" in w:
            assert abs(g["f_prev"] - w["f_prev"]) <= tol * max(1.0, abs(w["f_prev"]))


OPTIMIZERS = {"oLBFGS": oLBFGS_free, "SQN": SQN_free, "adaQN": adaQN_free}


# ------------------------------------------------------------------------------------------------
# lock-step driver: identical inputs into both libraries on EVERY call
# ------------------------------------------------------------------------------------------------
STATE_ARRAYS = ("gradient", "grad_prev", "x_sum", "x_avg_prev", "H0", "grad_sum_sq", "hess_vec")
MEM_ARRAYS = (("BFGS_mem", "s_mem"), ("BFGS_mem", "y_mem"), ("BFGS_mem", "s_bak"), ("BFGS_mem", "y_bak"),
              ("Fisher_mem", "F"))


def _arrays(opt):
    out = {}
    for name in STATE_ARRAYS:
        if hasattr(opt, name):
            out[name] = getattr(opt, name)
    for holder, name in MEM_ARRAYS:
        if hasattr(opt, holder):
            out[holder + "." + name] = getattr(getattr(opt, holder), name)
    return out


def run_lockstep(ref, opt, problem, x_ref, x_dev, step, ncalls, tol, on_sync=None, row_check=None):
    """Drive the oracle-backed `ref` and the library under test `opt` with the SAME inputs on every
    call: after each call all outputs and the complete optimiser state are compared (integers
    exactly, vectors norm-wise to `tol`), then the state of `opt` is overwritten with the oracle's,
    so rounding differences cannot be amplified by the (possibly ill-conditioned) trajectory."""
    last_grad_call = 999983     # a resumed run may start with a same-batch request
    for call in range(ncalls):
        rr = ref.run_optimizer(x_ref, step)
        ro = opt.run_optimizer(x_dev, step)
        where = "call %d (%s)" % (call, rr["task"])
        assert ro["task"] == rr["task"], where
        assert ro["info"] == rr["info"], where
        for k in ("niter", "section"):
            assert getattr(opt, k) == getattr(ref, k), (where, k)
        for k in ("mem_used", "mem_st_ix"):
            assert getattr(opt.BFGS_mem, k) == getattr(ref.BFGS_mem, k), (where, k)
            if hasattr(ref, "Fisher_mem"):
                assert getattr(opt.Fisher_mem, k) == getattr(ref.Fisher_mem, k), (where, "fisher " + k)
        if hasattr(ref, "f_prev"):
            assert opt.f_prev == ref.f_prev, where
        e = rel_err(to_np(x_dev), x_ref)
        assert e <= tol, "%s: x rel err %.3e" % (where, e)

        # what is requested, and where
        task = rr["task"]
        req_r, req_o = rr["requested_on"], ro["requested_on"]
        if task == "calc_hess_vec":
            assert _req_id(opt, x_dev, req_o[0]) == _req_id(ref, x_ref, req_r[0]), where
            for a, b_ in zip(req_o, req_r):
                e = rel_err(to_np(a), b_)
                assert e <= tol, "%s: request rel err %.3e" % (where, e)
        else:
            assert _req_id(opt, x_dev, req_o) == _req_id(ref, x_ref, req_r), where
            e = rel_err(to_np(req_o), req_r)
            assert e <= tol, "%s: request rel err %.3e" % (where, e)

        # complete state, array by array (pair / Fisher memories row by row)
        A_r, A_o = _arrays(ref), _arrays(opt)
        n = ref._n
        for name, a_r in A_r.items():
            a_o = to_np(A_o[name])
            a_r = np.asarray(a_r)
            if a_r.shape[0] > n and a_r.shape[0] % n == 0:
                for row in range(a_r.shape[0] // n):
                    # row_check(where, name, row, got_row, want_row, all_got, all_want) -> True when it has judged this row itself
                    if row_check and row_check(where, name, row, a_o[row * n:(row + 1) * n], a_r[row * n:(row + 1) * n], A_o, A_r):
                        continue
                    e = rel_err(a_o[row * n:(row + 1) * n], a_r[row * n:(row + 1) * n])
                    assert e <= tol, "%s: %s row %d rel err %.3e" % (where, name, row, e)
            else:
                e = rel_err(a_o, a_r)
                assert e <= tol, "%s: %s rel err %.3e" % (where, name, e)
        for name in ("buffer_rho", "buffer_alpha"):
            k = ref.BFGS_mem.mem_used
            a_r, a_o = getattr(ref.BFGS_mem, name)[:k], getattr(opt.BFGS_mem, name)[:k]
            if np.all(np.isfinite(a_r)) and rr["info"] == "no_problems_encountered":
                assert np.allclose(a_o, a_r, rtol=1e3 * tol, atol=1e-300), (where, name, a_o, a_r)

        # overwrite the state under test with the oracle's, then hand identical inputs to both
        opt._sp.assign(x_dev, x_ref)
        for name, a_r in A_r.items():
            opt._sp.assign(A_o[name], a_r)
        if on_sync:
            on_sync(opt)
        if task in ("calc_grad", "calc_grad_same_batch", "calc_grad_big_batch"):
            if task == "calc_grad":
                last_grad_call = call
            g = problem.grad(np.asarray(req_r).copy(), last_grad_call if task == "calc_grad_same_batch" else call)
            ref.update_gradient(g)
            opt.update_gradient(g)
        elif task == "calc_hess_vec":
            hv = problem.hess_vec(np.asarray(req_r[0]).copy(), np.asarray(req_r[1]).copy())
            ref.update_hess_vec(hv)
            opt.update_hess_vec(hv)
        elif task == "calc_fun_val_batch":
            f = problem.f(np.asarray(req_r).copy(), call)
            ref.update_function(f)
            opt.update_function(f)


class library_options:
    """`with library_options(lib, x_upload=0, register_host=1): ...` -- stochqn_hip_set_option for the block, the library's
    defaults (include/stochqn_hip.h) back afterwards."""
    DEFAULTS = {"x_upload": 1.0, "x_prefetch": 0.0, "register_host": 0.0, "register_min_bytes": float(4 << 20), "spec_x": 1.0,
                "apply_chunks": 8.0, "upload_slices": 8.0, "hash_threads": 0.0, "host_slice_min": float(1 << 21), "threepass": 1.0, "kappa_max": 1e6, "phase_ticks": 8000.0,
                "sdot_tile": 2.0, "fisher_tile": 2.0, "fisher_split": 1.0}

    def __init__(self, lib, **kw):
        import ctypes as C
        self.lib, self.kw = lib, kw
        lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
        for k in kw:
            assert k in self.DEFAULTS, k

    def __enter__(self):
        for k, v in self.kw.items():
            assert self.lib.stochqn_hip_set_option(k.encode(), float(v)) == 0, k
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            self.lib.stochqn_hip_set_option(k.encode(), self.DEFAULTS[k])
        return False


# the caller vouches: it keeps its hands off x while *req designates it, and its arrays outlive the optimiser
VOUCHED = dict(x_upload=0, x_prefetch=1, register_host=1)


def assert_no_host_range_left_pinned():
    """After a test of the host-caller path: every range that was page-locked through stochqn_hip_pin_host (stochqn_amd/free.py
    does that for the arrays it owns and for the user's x) has been unpinned by the time its array is gone, and the runtime
    refused none of those unpins.  A range that stays registered after its owner freed it is the recipe for a GPU memory fault at
    the next copy through that address."""
    import ctypes
    import gc
    import stochqn_amd
    gc.collect()
    for use_float in (False, True):
        h = stochqn_amd.cdll(use_float)
        h.stochqn_hip_stat.argtypes, h.stochqn_hip_stat.restype = [ctypes.c_char_p], ctypes.c_longlong
        assert h.stochqn_hip_stat(b"host_unpin_failed") == 0, "the runtime refused to unpin a host range"
        live = h.stochqn_hip_stat(b"host_pins_live")
        assert live == 0, "%d host range(s) still pinned after their test" % live
