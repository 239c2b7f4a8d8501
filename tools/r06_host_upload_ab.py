#!/usr/bin/env python3
"""Host callers (every array in host memory: reference src/Rwrapper.c:106-123, stochqn/pywrapper.pxi:161-172): which `x_upload`
should be the default?  VERDICT r05 #7: decide with data.

  x_upload = 1 (default): the caller's x goes up on every call that uses it (the reference's semantics: *req aliases x)
  x_upload = 2: nobody vouches for x; the library takes a checksum of all of the caller's x on host threads while the gradient
                travels and uploads x only when it differs from the device copy's

Interleaved A / B / A / B ..., SQN m = 5, L = 5 through the plain C ABI with structs rebuilt every call, at n = 1e8 (PCIe-bound:
0.8 GB per vector) and n = 1e6 (latency-bound), the five per-call arrays from stochqn_hip_alloc_host (pinned); and at n = 1e6 the
same with x, grad, hess_vec malloc'ed in the program-break heap (M_MMAP_THRESHOLD raised): the pinning rule DECLINES those, they
cross the link through the runtime's pageable path -- what that costs.  Seconds inside run_SQN only, per step, over two whole
L-cycles after two warm-up cycles.  One JSON line per (n, placement, x_upload, repetition) + a summary line.

    python tools/r06_host_upload_ab.py > gpurun_out/r06/host_upload_ab.jsonl
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import stochqn_amd
from stochqn_amd import _abi

lib = stochqn_amd.cdll()
be = stochqn_amd.lib()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_alloc_host.restype = C.c_void_p
lib.stochqn_hip_alloc_host.argtypes = [C.c_size_t, C.POINTER(C.c_int)]
lib.stochqn_hip_free_host.argtypes = [C.c_void_p, C.c_size_t]
lib.stochqn_hip_stat.argtypes = [C.c_char_p]
lib.stochqn_hip_stat.restype = C.c_longlong
libc = C.CDLL("libc.so.6")
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
M_MMAP_THRESHOLD = -3


def arrays(n, placement):
    """x, grad, hv, x_sum, x_avg_prev as numpy views + a function that gives the memory back"""
    ptrs, views, pinned = [], [], 0
    for _ in range(5):
        if placement == "alloc_host":
            flag = C.c_int(0)
            p = lib.stochqn_hip_alloc_host(8 * n, C.byref(flag))
            pinned += flag.value
        else:                                        # the program-break heap: malloc below the (raised) mmap threshold
            p = libc.malloc(8 * n)
        assert p
        ptrs.append(p)
        views.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (n,)))

    def free():
        for p in ptrs:
            if placement == "alloc_host":
                assert lib.stochqn_hip_free_host(p, 8 * n) == 0
            else:
                libc.free(p)
    return views, pinned, free


def one_run(n, placement, x_upload, m=5, L=5):
    lib.stochqn_hip_release_all()
    lib.stochqn_hip_stats_reset()
    assert lib.stochqn_hip_set_option(b"x_upload", float(x_upload)) == 0
    (x, grad, hv, x_sum, x_avg_prev), pinned, free = arrays(n, placement)
    rng = np.random.default_rng(5)
    d = 0.5 + rng.random(n)
    x[:] = 1.0 + rng.random(n)
    x_sum[:] = 0.0
    x_avg_prev[:] = 0.0
    S, Y = np.zeros(m * n), np.zeros(m * n)
    rho, alpha, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho.ctypes.data, alpha.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, m, 0, 0, L, 0.0, 0.0)
    w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, 0, 0, 1, 1, n)
    req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)
    view = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (n,))
    clock = [0.0]

    def step():
        target = w.niter + 1
        while w.niter < target or task.value != 101:
            if task.value == 104:
                np.multiply(d, view(req_vec.value), out=hv)
            else:
                np.multiply(d, view(req.value), out=grad)
            t0 = time.perf_counter()
            rc = be.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
            clock[0] += time.perf_counter() - t0
            assert rc in (0, 1), rc
    rc = be.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
    assert rc == 0
    for _ in range(2 * L):
        step()
    clock[0] = 0.0
    per = []
    for _ in range(2 * L):
        c0 = clock[0]
        step()
        per.append(1e3 * (clock[0] - c0))
    out = {"n": n, "placement": placement, "x_upload": x_upload, "arrays_pinned": pinned, "ms_per_step": round(sum(per) / len(per), 3),
           "ordinary_step_ms": round(sorted(per)[len(per) // 2], 3), "pairs_in_ring": int(b.mem_used),
           "x_uploads": int(lib.stochqn_hip_stat(b"x_uploads")), "x_uploads_skipped": int(lib.stochqn_hip_stat(b"x_uploads_skipped")),
           "pins_declined": int(lib.stochqn_hip_stat(b"host_pins_declined")), "f_end": float(0.5 * np.dot(d * x, x))}
    lib.stochqn_hip_release_all()
    free()
    return out


if __name__ == "__main__":
    reps = int(os.environ.get("AB_REPS", "3"))
    res = []
    for n, placement in ((100_000_000, "alloc_host"), (1_000_000, "alloc_host"), (1_000_000, "break_heap")):
        if placement == "break_heap":
            libc.mallopt(M_MMAP_THRESHOLD, 64 << 20)
        for rep in range(reps):
            for x_upload in (1, 2):
                r = dict(one_run(n, placement, x_upload), rep=rep)
                res.append(r)
                print(json.dumps(r), flush=True)
    lib.stochqn_hip_set_option(b"x_upload", 1.0)
    summary = {}
    for n, placement in ((100_000_000, "alloc_host"), (1_000_000, "alloc_host"), (1_000_000, "break_heap")):
        for x_upload in (1, 2):
            v = sorted(r["ms_per_step"] for r in res if (r["n"], r["placement"], r["x_upload"]) == (n, placement, x_upload))
            summary["n=%g %s x_upload=%d" % (n, placement, x_upload)] = {"median_ms_per_step": v[len(v) // 2], "all": v}
    for n, placement in ((100_000_000, "alloc_host"), (1_000_000, "alloc_host"), (1_000_000, "break_heap")):
        a = summary["n=%g %s x_upload=1" % (n, placement)]["median_ms_per_step"]
        b2 = summary["n=%g %s x_upload=2" % (n, placement)]["median_ms_per_step"]
        summary["n=%g %s: x_upload=2 over x_upload=1" % (n, placement)] = round(b2 / a, 4)
    print(json.dumps({"summary": summary}), flush=True)
