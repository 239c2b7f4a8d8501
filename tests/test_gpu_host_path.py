"""Host callers (R / numpy arrays crossing the ABI: reference src/Rwrapper.c:98-196, stochqn/pywrapper.pxi:161-207):
the PCIe-side machinery behind the unchanged ABI -- pinned caller arrays, x not re-uploaded when the device copy
is current, the update pass in slices with the download of x overlapping it -- must not change a single bit of
what a device-resident caller gets; and idle contexts that were moved to host memory under device-memory pressure
must come back as if nothing had happened."""
import ctypes as C

import numpy as np
import pytest

from harness import NoisyQuadratic, OPTIMIZERS, VOUCHED, compare_traces, library_options, own_mapping, rel_err, run_trace, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pins_go_with_their_arrays():
    yield
    from harness import assert_no_host_range_left_pinned
    assert_no_host_range_left_pinned()

TOL = 1e-10


def _lib():
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    lib.stochqn_hip_export.argtypes = [C.c_void_p]
    return lib


def stat(lib, name):
    v = lib.stochqn_hip_stat(name.encode())
    assert v >= 0, name
    return v


KW = {
    "oLBFGS": dict(mem_size=4, min_curvature=1e-4),
    "SQN": dict(mem_size=4, bfgs_upd_freq=3),
    "adaQN": dict(mem_size=4, fisher_size=5, bfgs_upd_freq=3, max_incr=1.01, rmsprop_weight=0.9),
}


@pytest.mark.parametrize("policy", ["default", "vouched", "checksum"])
@pytest.mark.parametrize("kind", ["oLBFGS", "SQN", "adaQN"])
@pytest.mark.parametrize("n", [1000, 2_500_001])
def test_host_and_device_callers_agree_bit_for_bit(kind, n, policy, hip_backend):
    """The same calls with the same inputs through numpy arrays and through torch tensors: every x, every request,
    every counter identical to the last bit.  n = 2,500,001 is past the thresholds of the host-side mechanisms (20 MB arrays
    are page-locked by their owner, stochqn_amd/free.py; the gradient comes up and x goes up and down in slices under the
    kernels) and odd, so every second ring row is off the 16-byte grid.  Three times: with the library's defaults (x goes up on
    every step, the library pins nothing by itself), for a caller that vouches for its arrays (x_upload = 0, register_host = 1),
    and with x_upload = 2, where nobody vouches and a checksum of all of x decides whether it goes up."""
    import torch
    lib = _lib()
    lib.stochqn_hip_stats_reset()
    P = NoisyQuadratic(n, seed=5)
    calls = 26
    with library_options(lib, **(VOUCHED if policy == "vouched" else (dict(x_upload=2) if policy == "checksum" else {}))):
        opt = OPTIMIZERS[kind](backend=hip_backend, space="host", **KW[kind])
        host = run_trace(opt, P, P.x0(), 0.05, calls)
        steps = sum(1 for t in host if t["changed"] == 1 or t["info"] == "search_direction_was_nan")
        opt.release()
    skipped, uploads, pinned = stat(lib, "x_uploads_skipped"), stat(lib, "x_uploads"), stat(lib, "host_ranges_registered")
    x = torch.as_tensor(P.x0(), device="cuda:0")
    dev = run_trace(OPTIMIZERS[kind](backend=hip_backend, space="device", device="cuda:0", **KW[kind]), P, x, 0.05, calls)
    for i, (h, d) in enumerate(zip(host, dev)):
        for k in ("task", "info", "changed", "niter", "section", "mem_used", "mem_st_ix", "req_id"):
            assert h[k] == d[k], (i, k)
        for k in ("x", "req", "req_vec"):
            if k in h:
                assert np.array_equal(h[k], d[k]), "call %d: %s differs between the host and the device caller" % (i, k)
    if policy == "vouched":
        # x goes up on the first step and again only after a request that was not at x (x_avg every L steps: the caller may
        # legally have touched x meanwhile); every other step reuses the device copy
        assert uploads >= 1 and skipped >= 5, (uploads, skipped)
        if kind == "oLBFGS":
            assert uploads == 1, uploads         # every request of oLBFGS is at x
    elif policy == "checksum" and n > 1_000_000:
        # the caller of this test never touches x: after the first step every ordinary step finds the sums equal
        assert uploads >= 1 and skipped >= steps - 4 >= 4, (uploads, skipped, steps)
    elif policy == "checksum":
        assert skipped == 0 and uploads >= steps                  # small x goes up in one piece, nothing to decide
    else:
        assert skipped == 0 and uploads >= steps >= 8, (uploads, skipped, steps)       # every step brings the caller's x up
    if n > 1_000_000:
        # the object's own arrays that cross the link (gradient; hess_vec / x_sum / x_avg_prev where the optimiser has them) have
        # mappings of their own and are page-locked by their owner; the test's x only where numpy did not put it in the break heap
        assert pinned >= (1 if kind == "oLBFGS" else 3), pinned
    else:
        assert pinned == 0                       # small arrays: not worth pinning
    lib.stochqn_hip_release_all()


@pytest.mark.parametrize("n,mode,edit", [(50_000, 1, "project"), (4_500_001, 1, "project"), (4_500_001, 2, "project"), (4_500_002, 2, "project"),
                                         (4_500_002, 2, "flip2"), (4_500_002, 2, "negate")])
def test_an_edit_of_one_coordinate_between_two_calls_moves_the_iterate(n, mode, edit, hip_backend, oracle_backend):
    """The reference's *req aliases x: a caller that clips or resets a FEW coordinates between two ordinary steps has simply
    moved the iterate.  256 probe values would miss such an edit; with the library's defaults (x_upload = 1) x goes up on
    every step -- for the large n in slices under the update -- and the trajectory equals the oracle's.  With x_upload = 2 a
    checksum of ALL of x decides: the steps after an edit send x up, the others do not -- same trajectory.  The edits: a
    projection of three coordinates; the SIGNS of two coordinates flipped; x -> -x with n even -- the last two are invisible to
    any sum of the words of x, plain or position-weighted (bit 63 twice cancels mod 2^64): the checksum mixes every
    word non-linearly with its position first (sqn_device.hpp: XHash)."""
    lib = _lib()
    lib.stochqn_hip_stats_reset()
    P = NoisyQuadratic(n, seed=13)
    kw = dict(mem_size=3, bfgs_upd_freq=4)
    where = [n // 3 + 1, n // 2 + 7, n - 2]           # none of them a probe position

    def drive(backend):
        opt = OPTIMIZERS["SQN"](backend=backend, space="host", **kw)
        x = P.x0()
        xs = []
        for call in range(22):
            r = opt.run_optimizer(x, 0.05)
            xs.append(x.copy())
            if r["task"] == "calc_hess_vec":
                rx, rv = r["requested_on"]
                opt.update_hess_vec(P.hess_vec(to_np(rx), to_np(rv)))
            else:
                if call in (6, 7, 13) and r["task"] == "calc_grad":
                    if edit == "project":
                        x[where] = 0.25                          # a projection of three coordinates, between two calls
                    elif edit == "flip2":
                        x[where[:2]] = -x[where[:2]]
                    else:
                        np.negative(x, out=x)
                opt.update_gradient(P.grad(to_np(r["requested_on"]), call))
        opt.release()
        return xs

    with library_options(lib, x_upload=mode):
        got = drive(hip_backend)
    uploads, skipped = stat(lib, "x_uploads"), stat(lib, "x_uploads_skipped")
    want = drive(oracle_backend)
    for i, (g, w) in enumerate(zip(got, want)):
        assert rel_err(g, w) <= TOL, i
        if edit == "project":
            assert np.array_equal(g[where] == 0.25, w[where] == 0.25), i
        else:
            assert np.array_equal(np.signbit(g[where]), np.signbit(w[where])), i
    if mode == 2:
        assert skipped >= 6 and 3 <= uploads <= 8, (uploads, skipped)      # the first step, the three edits, the calls that take x in one piece
    else:
        assert skipped == 0
    lib.stochqn_hip_release_all()


def test_the_library_pins_nothing_behind_the_callers_back(hip_backend):
    """Raw C-ABI callers (plain numpy arrays through ctypes, no binding that owns them): with the defaults no host range is
    registered by the library -- a caller may free and re-allocate its arrays between calls -- and arrays that the CALLER
    pins (stochqn_hip_pin_host) are used as pinned.  With register_host = 1 the library pins an array by itself, but only one
    it has seen at the same address in two consecutive calls."""
    from stochqn_amd import _abi
    lib = _lib()
    lib.stochqn_hip_pin_host.argtypes = [C.c_void_p, C.c_size_t]
    lib.stochqn_hip_unpin_host.argtypes = [C.c_void_p]
    import mmap
    n, m, L = 1_200_000, 3, 4
    d = 0.5 + np.random.default_rng(2).random(n)

    def own(count, fill=None):
        """An array with a mapping of its own (what malloc / numpy hand out above the mmap threshold, made deterministic here)."""
        a = np.frombuffer(mmap.mmap(-1, 8 * count), dtype=np.float64, count=count)
        if fill is not None:
            a[:] = fill
        return a

    def run(calls, fresh_arrays, pin):
        S, Y = np.zeros(m * n), np.zeros(m * n)
        x, grad, hv = own(n, 1.0 + np.random.default_rng(3).random(n)), own(n), own(n)
        x_sum, x_avg_prev, rho, alpha, dummy = own(n), own(n), np.zeros(m), np.zeros(m), np.zeros(1)
        if pin:
            for a in (x, grad, hv):
                assert lib.stochqn_hip_pin_host(a.ctypes.data, a.nbytes) == 0
        b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho.ctypes.data, alpha.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, m, 0, 0, L, 0.0, 0.0)
        w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, 0, 0, 1, 1, n)
        req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)
        view = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (n,))
        keep = []
        for _ in range(calls):
            if fresh_arrays:                                     # a caller that hands over a new gradient array every time
                grad = own(n)
                keep.append(grad)
            if task.value == 104:
                np.multiply(d, view(req_vec.value), out=hv)
            else:
                np.multiply(d, view(req.value), out=grad)
            rc = hip_backend.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
            assert rc in (0, 1)
        lib.stochqn_hip_release(C.c_void_p(S.ctypes.data))
        if pin:
            for a in (x, grad, hv):
                assert lib.stochqn_hip_unpin_host(a.ctypes.data) == 0
        return x

    lib.stochqn_hip_stats_reset()
    x_plain = run(14, False, False)
    assert stat(lib, "host_ranges_registered") == 0
    x_pinned = run(14, False, True)
    assert stat(lib, "host_ranges_registered") == 3              # the caller's three, through the API
    assert np.array_equal(x_plain, x_pinned)
    with library_options(lib, register_host=1, register_min_bytes=1 << 20):
        lib.stochqn_hip_stats_reset()
        run(14, True, False)
        fresh = stat(lib, "host_ranges_registered")
        lib.stochqn_hip_stats_reset()
        run(14, False, False)
        stable = stat(lib, "host_ranges_registered")
    assert stable >= 2 and fresh <= stable - 1, (fresh, stable)  # the ever-new gradient array is never pinned, x and the stable ones are
    # a block in the program-break heap is declined, whoever asks: it shares its first and last page with its neighbours and the
    # break moves under it (runtime.cpp: pinnable_in_place) -- and so is a range whose pages overlap a live pin
    declined0 = stat(lib, "host_pins_declined")
    small = np.zeros(6000)                                       # 48 KB: below every mmap threshold, so in the heap of the thread that made it
    assert lib.stochqn_hip_pin_host(small.ctypes.data, small.nbytes) == 1 and stat(lib, "host_pins_declined") == declined0 + 1
    assert lib.stochqn_hip_unpin_host(small.ctypes.data) == -1   # never was pinned
    # ... and so is a block of one of glibc's thread arenas: a numpy array a worker thread made, below the mmap threshold (which a
    # process that has freed large arrays has long since raised to its 32 MiB ceiling: the dynamic threshold).  What glibc itself
    # says about the block stands in the word before it: bit 1 = a mapping of its own, bit 2 = a heap of a thread's arena
    import threading
    made = []
    np.empty(24 << 20, dtype=np.uint8)                           # made and freed at once: the threshold is 24 MiB (or more) from here on
    worker = threading.Thread(target=lambda: made.extend(np.zeros(6 << 17) for _ in range(2)))      # 6 MiB each
    worker.start()
    worker.join()
    for a in made:
        head = C.c_size_t.from_address(a.ctypes.data - 8).value
        want = 0 if head & 2 else 1                              # mapped on its own -> pinned; inside a heap (bit 2: a thread arena's; neither: the break) -> declined
        got = lib.stochqn_hip_pin_host(a.ctypes.data, a.nbytes)
        said = "a worker thread's 6 MiB numpy array at %#x: glibc's chunk header %#x = %s; stochqn_hip_pin_host -> %d" % (
            a.ctypes.data, head, "a mapping of its own" if head & 2 else ("a thread arena's heap" if head & 4 else "the break heap"), got)
        print(said)
        import os
        try:                                                     # which case the box exercised, for the record of the run (gpurun_out/ is scratch)
            with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pin_rule_on_the_gpu_box.txt"), "a") as f:
                f.write(said + "\n")
        except OSError:
            pass
        assert got == want, said
        if want == 0:
            assert lib.stochqn_hip_unpin_host(a.ctypes.data) == 0
    big = own(2 * n)
    assert lib.stochqn_hip_pin_host(big.ctypes.data, 8 * n + 100) == 0
    assert lib.stochqn_hip_pin_host(big.ctypes.data + 8 * n + 200, 8 * n - 200) == 1      # starts in the page the first range ends in
    assert lib.stochqn_hip_pin_host(big.ctypes.data + 8 * n + 8192, 8 * n - 8192) == 0    # two pages on: pages of its own
    assert lib.stochqn_hip_unpin_host(big.ctypes.data) == 0 and lib.stochqn_hip_unpin_host(big.ctypes.data + 8 * n + 8192) == 0
    assert stat(lib, "host_pins_live") == 0
    lib.stochqn_hip_release_all()


def test_host_arrays_handed_out_by_the_library_are_pinned_and_carry_a_whole_run(hip_backend, oracle_backend):
    """stochqn_hip_alloc_host: the supported way for a caller that does not control its allocator (R, numpy, jemalloc ...) to get
    x / grad / hess_vec / x_sum / x_avg_prev that the pinning rule always accepts.  The arrays come back pinned, a 40-call SQN run
    over them (structs rebuilt on the stack every call, *req read on the host: reference src/Rwrapper.c:106-123) matches the oracle
    at 1e-10, and stochqn_hip_free_host takes every pin with it."""
    from stochqn_amd import _abi
    lib = _lib()
    lib.stochqn_hip_alloc_host.restype = C.c_void_p
    lib.stochqn_hip_alloc_host.argtypes = [C.c_size_t, C.POINTER(C.c_int)]
    lib.stochqn_hip_free_host.argtypes = [C.c_void_p, C.c_size_t]
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    n, m, L, calls = 3_000_001, 4, 3, 40
    live0 = lib.stochqn_hip_stat(b"host_pins_live")
    ptrs = []

    def lib_array(count):
        pinned = C.c_int(0)
        p = lib.stochqn_hip_alloc_host(8 * count, C.byref(pinned))
        assert p and pinned.value == 1
        ptrs.append((p, 8 * count))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (count,))

    d = 0.5 + np.random.default_rng(2).random(n)
    x0 = 1.0 + np.random.default_rng(3).random(n)

    def run(backend, alloc):
        S, Y = np.zeros(m * n), np.zeros(m * n)
        x, grad, hv, x_sum, x_avg_prev = (alloc(n) for _ in range(5))
        x[:] = x0
        rho, alpha, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
        b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho.ctypes.data, alpha.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, m, 0, 0, L, 0.0, 0.0)
        w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, 0, 0, 1, 1, n)
        req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)
        view = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (n,))
        log = []
        for _ in range(calls):
            if task.value == 104:
                np.multiply(d, view(req_vec.value), out=hv)
            else:
                np.multiply(d, view(req.value), out=grad)
            rc = backend.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
            assert rc in (0, 1)
            log.append((rc, task.value, info.value, w.niter, w.section, b.mem_used, b.mem_st_ix))
        out = x.copy()
        if backend is hip_backend:
            lib.stochqn_hip_release(C.c_void_p(S.ctypes.data))
        return out, log

    want, log_w = run(oracle_backend, lambda k: np.zeros(k))
    got, log_g = run(hip_backend, lib_array)
    assert log_g == log_w and rel_err(got, want) <= 1e-10
    assert lib.stochqn_hip_stat(b"host_pins_live") == live0 + 5
    for p, nbytes in ptrs:
        assert lib.stochqn_hip_free_host(p, nbytes) == 0
    assert lib.stochqn_hip_stat(b"host_pins_live") == live0


def test_x_edited_by_the_caller_between_calls_is_seen(hip_backend, oracle_backend):
    """The skipped upload must not turn x into the library's private variable: a caller that rescales / projects x between
    two calls gets its x used.  (The reference forbids touching *req -- include/stochqn.h:364-366 -- which is x after an
    ordinary step; the library still checks a spread of positions and uploads when anything moved.)"""
    lib = _lib()
    n = 50_000
    P = NoisyQuadratic(n, seed=9)
    kw = dict(mem_size=3, bfgs_upd_freq=4)

    def drive(backend):
        opt = OPTIMIZERS["SQN"](backend=backend, space="host", **kw)
        x = P.x0()
        xs = []
        for call in range(30):
            r = opt.run_optimizer(x, 0.05)
            xs.append(x.copy())
            if r["task"] == "calc_hess_vec":
                rx, rv = r["requested_on"]
                opt.update_hess_vec(P.hess_vec(to_np(rx), to_np(rv)))
            else:
                if call in (7, 16):
                    x *= 0.5                                     # the caller's own move, between two calls
                opt.update_gradient(P.grad(to_np(x if call in (7, 16) else r["requested_on"]), call))
        return xs

    lib.stochqn_hip_stats_reset()
    with library_options(lib, **VOUCHED):
        got = drive(hip_backend)
    want = drive(oracle_backend)
    for i, (g, w) in enumerate(zip(got, want)):
        assert rel_err(g, w) <= TOL, i
    assert stat(lib, "x_uploads") >= 3                           # the first step and the two edits
    lib.stochqn_hip_release_all()


def test_x_upload_is_the_default(hip_backend):
    lib = _lib()
    P = NoisyQuadratic(3000, seed=1)
    lib.stochqn_hip_stats_reset()
    run_trace(OPTIMIZERS["oLBFGS"](backend=hip_backend, space="host", mem_size=3), P, P.x0(), 0.05, 12)
    assert stat(lib, "x_uploads_skipped") == 0 and stat(lib, "x_uploads") >= 5
    with library_options(lib, x_upload=0):
        lib.stochqn_hip_stats_reset()
        run_trace(OPTIMIZERS["oLBFGS"](backend=hip_backend, space="host", mem_size=3), P, P.x0(), 0.05, 12)
        assert stat(lib, "x_uploads_skipped") >= 4 and stat(lib, "x_uploads") == 1
    lib.stochqn_hip_release_all()


@pytest.mark.parametrize("kind", ["SQN", "oLBFGS", "adaQN"])
def test_sliced_passes_equal_whole_launches(kind, hip_backend):
    """The two overlaps of the host path against their plain forms: pass 1 in slices that start as their part of the gradient
    lands (upload_slices; the accumulators are carried from launch to launch) and the update in slices whose x goes down
    while the next is computed (apply_chunks), against one launch each with the copies before / after: same bits.  A sliced
    pass 3 ends every slice by storing what its lanes have parked (clock-phased stores, `phase_ticks`): with the phases off and
    at a period of 300 ns the bits are the same again."""
    lib = _lib()
    n = 2_200_003
    P = NoisyQuadratic(n, seed=3)
    out = []
    try:
        for chunks, slices, ticks in ((1.0, 0.0, 8000.0), (8.0, 8.0, 8000.0), (3.0, 5.0, 8000.0), (1.0, 16.0, 8000.0), (8.0, 8.0, 0.0), (5.0, 3.0, 30.0)):
            assert lib.stochqn_hip_set_option(b"apply_chunks", chunks) == 0
            assert lib.stochqn_hip_set_option(b"upload_slices", slices) == 0
            assert lib.stochqn_hip_set_option(b"phase_ticks", ticks) == 0
            out.append(run_trace(OPTIMIZERS[kind](backend=hip_backend, space="host", **KW[kind]), P, P.x0(), 0.05, 16))
            lib.stochqn_hip_release_all()
    finally:
        lib.stochqn_hip_set_option(b"apply_chunks", 8.0)
        lib.stochqn_hip_set_option(b"upload_slices", 8.0)
        lib.stochqn_hip_set_option(b"phase_ticks", 8000.0)
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert a["task"] == b["task"] and a["info"] == b["info"]
            for k in ("x", "req", "req_vec"):
                if k in a:
                    assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("kind,strict,odd", [("SQN", 1, 1), ("oLBFGS", 0, 1), ("oLBFGS", 1, 0), ("adaQN", 0, 0)])
def test_x_sent_ahead_of_the_guard_leaves_the_same_bits(kind, strict, odd, hip_backend):
    """Option spec_x (default): pass 3 of the three-pass form in slices, each finished slice's x - step r on its way to the host
    before the guard has seen all of r, the guarded update under the transfer.  Against the plain host path (update, then the
    copies) and against a device-resident caller: the same bits in every x, request and counter -- also across a step that the
    guard REJECTS (NaN gradients around calls 9 and 17: the host array already holds the rejected step's x when the verdict
    comes, and must get the old x back).  With strict_grad the direction travels the same way (SQN / adaQN; oLBFGS hands
    back -step r, which the update itself writes: the plain path)."""
    import torch
    lib = _lib()
    n = 6_291_456 + odd                           # three rounds of pass 3's grid: the smallest shape that is sliced; odd: every other ring row off the 16-byte grid, one element beyond the last pack
    P = NoisyQuadratic(n, seed=11, nan_calls=(9, 10, 17))      # consecutive calls: one of them feeds a step whatever the call pattern
    calls = 24
    out, grads = {}, {}
    speculates = not (kind == "oLBFGS" and strict)
    try:
        assert lib.stochqn_hip_set_option(b"strict_grad", float(strict)) == 0     # 0 is the library's default (the suite runs with 1)
        for mode in (1.0, 0.0):
            assert lib.stochqn_hip_set_option(b"spec_x", mode) == 0
            lib.stochqn_hip_stats_reset()
            opt = OPTIMIZERS[kind](backend=hip_backend, space="host", **KW[kind])
            out[mode] = run_trace(opt, P, P.x0(), 0.05, calls)
            grads[mode] = np.array(opt.gradient, copy=True)
            ahead, again = stat(lib, "x_sent_ahead"), stat(lib, "x_sent_again")
            assert (ahead >= (2 if kind == "adaQN" else 6)) if (mode and speculates) else (ahead == 0), (mode, ahead)
            if mode and speculates and kind != "adaQN":
                assert again >= 1, "no step that went ahead was rejected: the test does not reach the path it is for"
            assert again <= ahead
            lib.stochqn_hip_release_all()
    finally:
        lib.stochqn_hip_set_option(b"spec_x", 1.0)
        lib.stochqn_hip_set_option(b"strict_grad", 1.0)
    assert np.array_equal(grads[1.0], grads[0.0], equal_nan=True), "the caller's gradient array differs between the two host paths"
    x = torch.as_tensor(P.x0(), device="cuda:0")
    dev = run_trace(OPTIMIZERS[kind](backend=hip_backend, space="device", device="cuda:0", **KW[kind]), P, x, 0.05, calls)
    assert any(t["info"] == "search_direction_was_nan" for t in dev), "the NaN gradient was meant to get a step rejected"
    for name, other in (("plain host path", out[0.0]), ("device caller", dev)):
        for i, (a, b) in enumerate(zip(out[1.0], other)):
            for k in ("task", "info", "changed", "niter", "section", "mem_used", "mem_st_ix", "req_id"):
                assert a[k] == b[k], (name, i, k)
            for k in ("x", "req", "req_vec"):
                if k in a:
                    assert np.array_equal(a[k], b[k], equal_nan=True), "call %d: %s differs from the %s" % (i, k, name)
    lib.stochqn_hip_release_all()


@pytest.mark.parametrize("kind", ["SQN", "adaQN"])
def test_x_sent_up_while_the_caller_computes_changes_nothing(kind, hip_backend):
    """Option x_prefetch (with x_upload = 0; both opt-in): a call that returns with *req == x while the device copy of x is stale (the request before was
    at x_avg: Hessian-vector product, big-batch gradient, function value) starts the upload of x on a side stream and returns;
    the copy runs while the caller evaluates its gradient, the next call orders itself behind it.  Same bits as with the upload
    inside the next call -- also when the caller, against the contract, edits x while it is on its way (the probe values catch
    that and x goes up again)."""
    lib = _lib()
    n = 2_500_001
    P = NoisyQuadratic(n, seed=9)
    EDITS = (-1, 14, 15)                          # no edit; an edit right after one of two consecutive calls: one of them follows a request at x_avg

    def drive(edit_at):
        opt = OPTIMIZERS[kind](backend=hip_backend, space="host", **KW[kind])
        x = own_mapping(P.x0())                      # a prefetch needs a page-locked x: pages of its own, whatever numpy's allocator does
        xs, last = [], 999983
        for call in range(30):
            r = opt.run_optimizer(x, 0.05)
            xs.append(x.copy())
            task = r["task"]
            if task == "calc_hess_vec":
                rx, rv = r["requested_on"]
                opt.update_hess_vec(P.hess_vec(to_np(rx), to_np(rv)))
            elif task == "calc_fun_val_batch":
                opt.update_function(P.f(to_np(r["requested_on"]), call))
            else:
                if call == edit_at and task == "calc_grad":
                    x *= 0.5                                 # while the prefetched copy may still be in flight
                if task == "calc_grad":
                    last = call
                opt.update_gradient(P.grad(to_np(x if call == edit_at else r["requested_on"]), last if task == "calc_grad_same_batch" else call))
        opt.release()
        return xs

    out = {}
    for mode in (1.0, 0.0):
        with library_options(lib, x_upload=0, register_host=1, x_prefetch=mode):
            for edit_at in EDITS:
                lib.stochqn_hip_stats_reset()
                out[mode, edit_at] = drive(edit_at)
                pre = stat(lib, "x_prefetched")
                assert (pre >= 2) if mode else (pre == 0), (mode, edit_at, pre)
                lib.stochqn_hip_release_all()
    for edit_at in EDITS:
        for i, (a, b) in enumerate(zip(out[1.0, edit_at], out[0.0, edit_at])):
            assert np.array_equal(a, b), "x after call %d differs (caller's edit at call %d)" % (i, edit_at)


def test_the_event_profiler_nests_with_the_sliced_host_path(hip_backend):
    """STOCHQN_HIP_PROFILE / stochqn_hip_profile_enable with a HOST caller of a size whose passes run in slices: a sliced pass hands
    every finished slice to a callback that launches kernels of its own INSIDE the pass's profiling scope.  Until round 5 the inner
    scope's end closed the outer pair, the outer pair's second event was never recorded, and the next synchronisation failed the
    call (-1000, "invalid resource handle") -- found by tools/bench_configs.py c3host.  Now: every call succeeds, the kernels of
    the sliced passes are all counted, and the trajectory is the unprofiled one bit for bit."""
    lib = _lib()
    lib.stochqn_hip_profile_get.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
    lib.stochqn_hip_profile_name.restype = C.c_char_p
    n = 6_291_457
    P = NoisyQuadratic(n, seed=21)
    plain = run_trace(OPTIMIZERS["SQN"](backend=hip_backend, space="host", **KW["SQN"]), P, P.x0(), 0.05, 14)
    lib.stochqn_hip_release_all()
    try:
        lib.stochqn_hip_profile_enable(1)
        lib.stochqn_hip_profile_reset()
        lib.stochqn_hip_stats_reset()
        prof = run_trace(OPTIMIZERS["SQN"](backend=hip_backend, space="host", **KW["SQN"]), P, P.x0(), 0.05, 14)     # raises on -1000
        lib.stochqn_hip_release_all()
        seen = {}
        for i in range(lib.stochqn_hip_profile_kernels()):
            cnt, ms = C.c_longlong(), C.c_double()
            lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
            if cnt.value:
                seen[lib.stochqn_hip_profile_name(i).decode()] = cnt.value
    finally:
        lib.stochqn_hip_profile_enable(0)
    assert stat(lib, "x_sent_ahead") >= 3                                   # the path under test ran: pass 3 in slices, x ahead of the guard
    assert seen.get("sadd", 0) >= 3 and seen.get("apply", 0) >= 3 and seen.get("sdot", 0) + seen.get("sdot2", 0) >= 3, seen
    for a, b in zip(plain, prof):
        assert a["task"] == b["task"] and a["info"] == b["info"] and np.array_equal(a["x"], b["x"])
    lib.stochqn_hip_release_all()


def test_step_counters_name_the_form_that_ran(hip_backend):
    """stochqn_hip_stat: which form of the recursion each step took."""
    import torch
    lib = _lib()
    P = NoisyQuadratic(4000, seed=2)
    lib.stochqn_hip_stats_reset()
    x = torch.as_tensor(P.x0(), device="cuda:0")
    run_trace(OPTIMIZERS["oLBFGS"](backend=hip_backend, space="device", device="cuda:0", mem_size=3), P, x, 0.05, 21)
    plain, three = stat(lib, "steps_plain"), stat(lib, "steps_three_pass")
    assert plain == 1 and three == 9 and stat(lib, "steps_sweeps") == 0, (plain, three)
    try:
        lib.stochqn_hip_set_option(b"threepass", 0.0)
        lib.stochqn_hip_stats_reset()
        x = torch.as_tensor(P.x0(), device="cuda:0")
        run_trace(OPTIMIZERS["oLBFGS"](backend=hip_backend, space="device", device="cuda:0", mem_size=3), P, x, 0.05, 21)
        assert stat(lib, "steps_sweeps") == 9 and stat(lib, "steps_three_pass") == 0 and stat(lib, "steps_kappa_fallback") == 0
    finally:
        lib.stochqn_hip_set_option(b"threepass", 1.0)
    assert lib.stochqn_hip_stat(b"no_such_counter") == -1
    lib.stochqn_hip_release_all()


# ------------------------------------------------------------------------------------------------
# idle contexts under device-memory pressure
# ------------------------------------------------------------------------------------------------
def _advance(opt, P, x, ncalls, first_call):
    last = getattr(opt, "_last_grad_call", 999983)           # a same-batch request reuses the noise of the calc_grad before it
    for call in range(first_call, first_call + ncalls):
        r = opt.run_optimizer(x, 0.05)
        if r["task"] == "calc_hess_vec":
            rx, rv = r["requested_on"]
            opt.update_hess_vec(P.hess_vec(to_np(rx), to_np(rv)))
        elif r["task"] == "calc_fun_val_batch":
            opt.update_function(P.f(to_np(r["requested_on"]), call))
        else:
            if r["task"] == "calc_grad":
                last = opt._last_grad_call = call
            opt.update_gradient(P.grad(to_np(r["requested_on"]), last if r["task"] == "calc_grad_same_batch" else call))


@pytest.mark.parametrize("kind", ["oLBFGS", "SQN", "adaQN"])
def test_abandoned_host_optimisers_are_reclaimed_and_a_survivor_resumes(kind, hip_backend, oracle_backend):
    """R and Python never call dealloc_*.  With the mirrors capped (option max_mirror_bytes) a stream of host-array optimisers
    that are used for a few steps and dropped keeps running: the least recently used idle contexts are moved to host
    memory.  One object is kept alive and left idle meanwhile; when it is called again it continues from the library's
    own copy of its state -- its own numpy arrays have been stale since its first call -- exactly like an oracle run
    that was never interrupted."""
    lib = _lib()
    n = 40_000
    P = NoisyQuadratic(n, seed=11)
    kw = KW[kind]
    ref = OPTIMIZERS[kind](backend=oracle_backend, space="host", **kw)
    x_ref = P.x0()
    _advance(ref, P, x_ref, 40, 0)

    try:
        per_ctx = (2 * kw["mem_size"] + 8) * n * 8
        assert lib.stochqn_hip_set_option(b"max_mirror_bytes", float(3 * per_ctx)) == 0
        lib.stochqn_hip_stats_reset()
        keep = OPTIMIZERS[kind](backend=hip_backend, space="host", **kw)
        x = P.x0()
        _advance(keep, P, x, 17, 0)
        idle = []
        for j in range(12):                                  # used for a few steps, then never again, never released: what a
            other = OPTIMIZERS[kind](backend=hip_backend, space="host", **kw)     # long R session leaves behind
            xo = P.x0()
            _advance(other, P, xo, 9, 0)
            idle.append((other, xo))                         # (kept referenced so that every object has arrays -- a key -- of its own)
        reclaimed = stat(lib, "contexts_reclaimed")
        assert reclaimed >= 8, reclaimed
        _advance(keep, P, x, 23, 17)                         # comes back from the spill
        assert keep.niter == ref.niter and keep.BFGS_mem.mem_used == ref.BFGS_mem.mem_used
        assert rel_err(x, x_ref) <= 1e-9
        # checkpoint of an object whose context is spilled: export brings its own arrays up to date
        more = OPTIMIZERS[kind](backend=hip_backend, space="host", **kw)
        xm = P.x0()
        _advance(more, P, xm, 9, 0)
        for j in range(4):
            o2 = OPTIMIZERS[kind](backend=hip_backend, space="host", **kw)
            xo = P.x0()
            _advance(o2, P, xo, 9, 0)
            idle.append((o2, xo))
        assert lib.stochqn_hip_export(C.c_void_p(more.BFGS_mem.s_mem.ctypes.data)) == 0
        assert np.any(more.BFGS_mem.s_mem != 0)
    finally:
        lib.stochqn_hip_set_option(b"max_mirror_bytes", 0.0)
        lib.stochqn_hip_release_all()


def test_running_out_of_device_memory_reclaims_instead_of_failing(hip_backend, oracle_backend):
    """The real thing: hipMalloc fails because the device is full; the idle context goes to host memory and the call succeeds."""
    import torch
    lib = _lib()
    n, m = 6_000_000, 8                                      # S + Y mirrors: 2 * 8 * 6e6 * 8 = 768 MB per optimiser
    P = NoisyQuadratic(n, seed=4)
    kw = dict(mem_size=m, bfgs_upd_freq=2)
    ref = OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw)
    x_ref = P.x0()
    _advance(ref, P, x_ref, 12, 0)
    lib.stochqn_hip_stats_reset()
    a = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
    xa = P.x0()
    _advance(a, P, xa, 6, 0)
    # a few more streams in use before the device is filled (cheap insurance; streams first used with 300 MB left work too:
    # profiles/src/queue_oom.hip)
    warm = [torch.cuda.Stream() for _ in range(8)]
    for s in warm:
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda:0").add_(1)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    ballast = torch.empty(free - (700 << 20), dtype=torch.uint8, device="cuda:0")     # leaves less than a second optimiser needs
    try:
        b = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
        xb = P.x0()
        _advance(b, P, xb, 6, 0)                             # its mirrors only fit once a's have been moved out
        assert stat(lib, "contexts_reclaimed") >= 1
        _advance(a, P, xa, 6, 6)                             # and a comes back (b is idle now and makes room in turn)
        assert stat(lib, "contexts_reclaimed") >= 2
        assert a.niter == ref.niter and rel_err(xa, x_ref) <= 1e-9
    finally:
        del ballast
        torch.cuda.empty_cache()
        lib.stochqn_hip_release_all()



def test_function_increase_rolls_x_back_for_host_callers_too(hip_backend):
    """adaQN's `func_increased` branch writes x itself (x <- x_avg_prev, reference src/stochqn.c:1275-1283) in a call that
    takes no step: the device copy, the host array and the 'x is current' bookkeeping must all follow -- host and device
    callers bit for bit, with a spike in f forcing the branch."""
    import torch
    lib = _lib()
    n = 2_300_000
    P = NoisyQuadratic(n, seed=6, f_spike_calls=range(12, 15))
    kw = dict(mem_size=3, fisher_size=4, bfgs_upd_freq=3, max_incr=1.01, rmsprop_weight=0.9)
    host = run_trace(OPTIMIZERS["adaQN"](backend=hip_backend, space="host", **kw), P, P.x0(), 0.02, 30)
    x = torch.as_tensor(P.x0(), device="cuda:0")
    dev = run_trace(OPTIMIZERS["adaQN"](backend=hip_backend, space="device", device="cuda:0", **kw), P, x, 0.02, 30)
    assert any(r["info"] == "func_increased" for r in host)
    for i, (h, d) in enumerate(zip(host, dev)):
        assert h["task"] == d["task"] and h["info"] == d["info"] and h["niter"] == d["niter"], i
        assert np.array_equal(h["x"], d["x"]) and np.array_equal(h["req"], d["req"]), i
    lib.stochqn_hip_release_all()


@pytest.mark.parametrize("kind,x_upload", [("SQN", 1), ("adaQN", 1), ("SQN", 2)])
def test_float_host_and_device_callers_agree_bit_for_bit(kind, x_upload):
    """The same for the single-precision library (libstochqn_f32.so): pinned float arrays, sliced pass 1, sliced pass 3 with x on
    its way ahead of the guard (packs of four floats: n = 9,000,003 leaves three elements beyond the last pack and is two
    rounds of pass 3's grid).  With x_upload = 2 the checksum of x has a ragged end in both places -- 36,000,012 bytes are a
    half 64-bit word beyond the last whole one for the host's threads and 12 bytes beyond the last 16-byte pair for the kernel
    -- and uploads are only ever skipped if the two agree on it."""
    import stochqn_amd
    import torch
    be = stochqn_amd.lib(use_float=True)
    lib = stochqn_amd.cdll(use_float=True)
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    lib.stochqn_hip_stats_reset()
    n = 9_000_003
    P = NoisyQuadratic(n, seed=5)
    kw = dict(KW[kind], use_float=True)
    x0 = P.x0().astype(np.float32)
    with library_options(lib, x_upload=x_upload):
        host = run_trace(OPTIMIZERS[kind](backend=be, space="host", **kw), P, x0.copy(), 0.05, 16)
    if x_upload == 2:
        assert lib.stochqn_hip_stat(b"x_uploads_skipped") >= 4, (lib.stochqn_hip_stat(b"x_uploads_skipped"), lib.stochqn_hip_stat(b"x_uploads"))
    x = torch.as_tensor(x0.copy(), device="cuda:0")
    dev = run_trace(OPTIMIZERS[kind](backend=be, space="device", device="cuda:0", **kw), P, x, 0.05, 16)
    for i, (h, d) in enumerate(zip(host, dev)):
        assert h["task"] == d["task"] and h["info"] == d["info"], i
        for k in ("x", "req", "req_vec"):
            if k in h:
                assert np.array_equal(h[k], d[k]), (i, k)
    assert lib.stochqn_hip_stat(b"x_sent_ahead") >= 3, lib.stochqn_hip_stat(b"x_sent_ahead")
    lib.stochqn_hip_release_all()
