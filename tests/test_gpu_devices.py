"""Single-process multi-device mode (stochqn_amd/csrc/group.cpp) on ONE GPU: option "devices" = P with
"virtual_devices" = 1 puts P shards -- each with its own host thread, device context and stream -- on the
one physical device and sums their partial dot products through the host-side rendezvous reducer (RCCL
refuses two ranks on one device).  Everything else is the code an 8-GPU node runs: the unchanged reference
ABI called by ONE host process, n cut into contiguous slices, the fan-out of every run_* call, the
scatter of x / grad / hess_vec and the gather of x / direction / *req / *req_vec, identical decisions on
all shards.  Bar: the unsharded CPU oracle, integers exactly, vectors to 1e-10."""
import ctypes as C

import numpy as np
import pytest

from harness import own_mapping, OPTIMIZERS, NoisyQuadratic, compare_traces, rel_err, run_trace
from test_gpu_parity import CONFIGS, FREE_RUN_TOL, TOL
from test_oracle_known_answers import GOLD, host_view, run_c_rosen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pins_go_with_their_arrays():
    yield
    from harness import assert_no_host_range_left_pinned
    assert_no_host_range_left_pinned()

LOOP, RCCL = 3, 1           # Reducer::Kind of runtime.hpp


def _lib():
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    lib.stochqn_hip_devices_active.argtypes = [C.c_void_p]
    lib.stochqn_hip_devices_reducer.argtypes = [C.c_void_p]
    lib.stochqn_hip_export.argtypes = [C.c_void_p]
    lib.stochqn_hip_release.argtypes = [C.c_void_p]
    return lib


@pytest.fixture(params=[2, 3])
def devices(request, hip_backend):
    lib = _lib()
    assert lib.stochqn_hip_set_option(b"virtual_devices", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices", float(request.param)) == 0
    yield request.param
    lib.stochqn_hip_release_all()
    lib.stochqn_hip_set_option(b"devices", 0.0)
    lib.stochqn_hip_set_option(b"virtual_devices", 0.0)
    lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


NAMES = ["olbfgs_default", "olbfgs_nocurv_hess_init", "olbfgs_nan_grad", "olbfgs_reject_all", "sqn_hessvec", "sqn_graddiff",
         "sqn_reject", "sqn_nan", "sqn_nonan_check", "adaqn_fisher_rms", "adaqn_graddiff", "adaqn_func_increased", "adaqn_nan",
         "adaqn_fisher500", "sqn_ring20", "sqn_ring30", "sqn_ring50"]


@pytest.mark.parametrize("n", [3001, 64])
@pytest.mark.parametrize("name", NAMES)
def test_host_caller_on_P_devices_equals_unsharded_oracle(name, n, devices, hip_backend, oracle_backend):
    """Profile B (the R / Cython protocol): numpy arrays, structs rebuilt per call, ONE process.  The
    trace -- every task, info, counter, which array *req aliases, x, *req and *req_vec after every call --
    must equal the oracle's on the unsharded problem."""
    lib = _lib()
    cfg = [c for c in CONFIGS if c[0] == name][0]
    _, optname, kw, step, calls, pkw = cfg
    P = NoisyQuadratic(n, seed=7, **pkw)
    want = run_trace(OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw), P, P.x0(), step, calls)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
    x = P.x0()
    got = run_trace(opt, P, x, step, calls)
    key = C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)
    assert lib.stochqn_hip_devices_active(key) == devices
    assert lib.stochqn_hip_devices_reducer(key) == LOOP
    compare_traces(got, want, FREE_RUN_TOL.get(name, TOL))
    opt.release()


@pytest.fixture
def one_shard_over_rccl(hip_backend):
    """Option devices_rccl_single: the multi-device mode with ONE shard on the current device over a REAL communicator --
    ncclCommInitAll of one device, the shard's worker thread, ncclAllReduce on the shard's stream for every reduction,
    ncclCommDestroy -- i.e. every line of group.cpp / runtime.cpp that an 8-GPU node runs and a one-GPU box can."""
    lib = _lib()
    assert lib.stochqn_hip_set_option(b"devices", 0.0) == 0 and lib.stochqn_hip_set_option(b"virtual_devices", 0.0) == 0
    assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices_rccl_single", 1.0) == 0
    yield 1
    lib.stochqn_hip_release_all()
    lib.stochqn_hip_set_option(b"devices_rccl_single", 0.0)
    lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


@pytest.mark.parametrize("name", ["olbfgs_default", "olbfgs_nan_grad", "sqn_hessvec", "sqn_graddiff", "sqn_reject", "adaqn_fisher_rms",
                                  "adaqn_func_increased", "sqn_ring20"])
def test_one_shard_over_a_real_rccl_communicator(name, one_shard_over_rccl, hip_backend, oracle_backend):
    """SQN / oLBFGS / adaQN through the group mode's RCCL path on one device, against the oracle: every reduction of every
    step is a real ncclAllReduce on the communicator that ncclCommInitAll made."""
    lib = _lib()
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    cfg = [c for c in CONFIGS if c[0] == name][0]
    _, optname, kw, step, calls, pkw = cfg
    n = 3001
    P = NoisyQuadratic(n, seed=7, **pkw)
    want = run_trace(OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw), P, P.x0(), step, calls)
    lib.stochqn_hip_stats_reset()
    opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
    got = run_trace(opt, P, P.x0(), step, calls)
    key = C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)
    assert lib.stochqn_hip_devices_active(key) == 1 and lib.stochqn_hip_devices_reducer(key) == RCCL
    assert lib.stochqn_hip_stat(b"allreduces") >= calls // 2
    compare_traces(got, want, FREE_RUN_TOL.get(name, TOL))
    opt.release()


def test_library_owned_workspace_on_one_shard_over_rccl(one_shard_over_rccl, hip_backend, oracle_backend):
    test_library_owned_sharded_workspaces("SQN-hessvec", 1, hip_backend, oracle_backend)
    test_library_owned_sharded_workspaces("adaQN-fisher", 1, hip_backend, oracle_backend)


@pytest.mark.parametrize("kind,kw", [("SQN", dict(mem_size=3, bfgs_upd_freq=3)), ("oLBFGS", dict(mem_size=3)),
                                     ("adaQN", dict(mem_size=3, fisher_size=5, bfgs_upd_freq=3, max_incr=1.01, rmsprop_weight=0.9))])
def test_host_path_of_the_multi_device_mode(kind, kw, devices, hip_backend, oracle_backend):
    """The host-caller machinery of one device applies to the shards too: the caller's x / grad / hess_vec are pinned once
    for all devices (20 MB arrays here), and x goes up only when the shards' copies may be out of date -- and the caller that
    edits x between two calls is still seen.  Bar: the unsharded oracle."""
    lib = _lib()
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    n = 2_500_001
    P = NoisyQuadratic(n, seed=5)

    def drive(backend):
        opt = OPTIMIZERS[kind](backend=backend, space="host", **kw)
        x = P.x0()
        xs, last, edited = [], 999983, False
        for call in range(18):
            r = opt.run_optimizer(x, 0.05)
            xs.append(x.copy())
            if r["task"] == "calc_hess_vec":
                rx, rv = r["requested_on"]
                opt.update_hess_vec(P.hess_vec(np.asarray(rx).copy(), np.asarray(rv).copy()))
            elif r["task"] == "calc_fun_val_batch":
                opt.update_function(P.f(np.asarray(r["requested_on"]).copy(), call))
            else:
                edit = call >= 10 and r["task"] == "calc_grad" and not edited
                if edit:
                    x *= 0.75                                # the caller's own move between two calls (once)
                    edited = True
                if r["task"] == "calc_grad":
                    last = call
                at = x if edit else r["requested_on"]
                opt.update_gradient(P.grad(np.asarray(at).copy(), last if r["task"] == "calc_grad_same_batch" else call))
        return xs, opt

    from harness import VOUCHED, library_options
    lib.stochqn_hip_stats_reset()
    with library_options(lib, **VOUCHED):                   # a caller that vouches for its arrays: skipped uploads, pinned by the library
        got, opt = drive(hip_backend)
    assert lib.stochqn_hip_devices_active(C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)) == devices
    skipped, uploads, pinned = (lib.stochqn_hip_stat(k) for k in (b"x_uploads_skipped", b"x_uploads", b"host_ranges_registered"))
    want, _ = drive(oracle_backend)
    for i, (g, w) in enumerate(zip(got, want)):
        assert rel_err(g, w) <= 1e-9, i
    # pinned: at least the object's own gradient array (a mapping of its own, stochqn_amd/free.py); the test's x only if numpy did not put it in the break heap
    assert skipped >= 3 and uploads >= 2 and pinned >= 1, (skipped, uploads, pinned)
    opt.release()


def test_large_shards_run_the_overlaps_of_the_host_path(hip_backend, oracle_backend):
    """Shards large enough for every overlap of the single-device host path (each shard's machine is handed its slice of the
    caller's arrays as host pointers): pass 1 in slices under the upload, x sent ahead of the guard, x sent up in the background
    after a request at x_avg -- on three shards, n odd.  Bar: the unsharded oracle, with a NaN gradient in the middle (a step
    that went ahead is rejected on every shard)."""
    lib = _lib()
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    assert lib.stochqn_hip_set_option(b"virtual_devices", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices", 3.0) == 0
    n = 13_000_001                                      # 4.33e6 per shard: the smallest shard whose pass 3 runs in slices (two rounds of its grid)
    P = NoisyQuadratic(n, seed=8, nan_calls=(10, 11))
    kw = dict(mem_size=3, bfgs_upd_freq=3)
    from harness import VOUCHED, library_options
    try:
        assert lib.stochqn_hip_set_option(b"strict_grad", 0.0) == 0
        want = run_trace(OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw), P, P.x0(), 0.05, 20)
        # the defaults (x comes up with the slices of the update), the vouching caller, and x_upload = 2 (each shard takes the
        # checksum of its slice of the caller's x)
        for policy in ({}, VOUCHED, dict(x_upload=2)):
            with library_options(lib, **policy):
                lib.stochqn_hip_stats_reset()
                opt = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
                # x in a mapping of its own: numpy may well put a 104 MB array INTO the break heap (malloc serves a request from a free
                # chunk of the heap before it thinks of mmap, whatever the mmap threshold says -- seen in round 5: a 104 MB x at
                # 0x61d38a78ab70 inside a 700 MB heap), where the library declines to page-lock it and nothing can be sent ahead
                got = run_trace(opt, P, own_mapping(P.x0()), 0.05, 20)
                assert lib.stochqn_hip_devices_active(C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)) == 3
                ahead, again, pre = (lib.stochqn_hip_stat(k) for k in (b"x_sent_ahead", b"x_sent_again", b"x_prefetched"))
                pins = {k.decode(): lib.stochqn_hip_stat(k) for k in (b"host_ranges_registered", b"host_pins_declined", b"host_pins_foreign", b"host_pin_errors", b"host_pins_live")}
                assert ahead >= 3 * 4 and again >= 3 and (pre >= 3 if policy is VOUCHED else pre == 0), (ahead, again, pre, pins)      # per shard
                if policy.get("x_upload") == 2:
                    assert lib.stochqn_hip_stat(b"x_uploads_skipped") >= 3 * 4, lib.stochqn_hip_stat(b"x_uploads_skipped")
                assert any(t["info"] == "search_direction_was_nan" for t in want)
                compare_traces(got, want, 1e-9)
                opt.release()
    finally:
        lib.stochqn_hip_set_option(b"strict_grad", 1.0)
        lib.stochqn_hip_release_all()
        lib.stochqn_hip_set_option(b"devices", 0.0)
        lib.stochqn_hip_set_option(b"virtual_devices", 0.0)
        lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


def test_c_rosen_protocol_on_P_devices(devices, hip_backend):
    """Profile A: initialize_SQN / run_SQN / dealloc_SQN exactly as reference example/c_rosen.c:100-125 does,
    the workspace sharded over the devices (n = 4: shards of 1-2 variables), *req and *req_vec read on the host."""
    k = GOLD["c_rosen"]
    out = run_c_rosen(hip_backend, np.array(k["x0"]), host_view)
    for key in ("f_initial", "f_it10", "f_it200", "f_final"):
        assert out[key].strip() == k[key], (key, out[key])
    assert out["x_final"] == k["x_final"]


@pytest.mark.parametrize("which", ["oLBFGS", "SQN-hessvec", "SQN-graddiff", "adaQN-fisher", "adaQN-graddiff"])
def test_library_owned_sharded_workspaces(which, devices, hip_backend, oracle_backend):
    """initialize_* in group mode hands out a workspace whose arrays exist only as device slices; a plain
    host caller (malloc'ed x / grad, reads *req on the host) must see the oracle's run."""
    lib = _lib()
    n = 20011
    P = NoisyQuadratic(n, seed=4, f_spike_calls=range(30, 36))

    def drive(be):
        x, g, hv = P.x0(), np.zeros(n), np.zeros(n)
        req, rv, task, info = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        if which == "oLBFGS":
            w = be.initialize_oLBFGS(n, 4, 0.0, 0.0, 1e-4, 1, 1)
            run = lambda f: be.run_oLBFGS(0.1, x.ctypes.data, g.ctypes.data, C.byref(req), C.byref(task), w, C.byref(info))
            free = be.dealloc_oLBFGS
        elif which.startswith("SQN"):
            w = be.initialize_SQN(n, 3, 4, 1e-4, int(which.endswith("graddiff")), 0.0, 1, 1)
            run = lambda f: be.run_SQN(0.1, x.ctypes.data, g.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(rv), C.byref(task), w, C.byref(info))
            free = be.dealloc_SQN
        else:
            w = be.initialize_adaQN(n, 3, 6, 4, 1.01, 1e-4, 1e-4, 0.9, int(which.endswith("graddiff")), 0.0, 1, 1)
            run = lambda f: be.run_adaQN(0.05, x.ctypes.data, f, g.ctypes.data, C.byref(req), C.byref(task), w, C.byref(info))
            free = be.dealloc_adaQN
        assert bool(w), "initialize returned NULL"
        if be is hip_backend:
            assert lib.stochqn_hip_devices_active(C.c_void_p(w.contents.bfgs_memory.contents.s_mem)) == devices
        out, f, last = [], 0.0, 0
        for call in range(60):
            rc = run(f)
            assert rc in (0, 1), rc
            at = host_view(req.value, n).copy()
            rec = [rc, task.value, info.value, w.contents.niter, w.contents.section,
                   w.contents.bfgs_memory.contents.mem_used, w.contents.bfgs_memory.contents.mem_st_ix, x.copy(), at]
            if task.value in (101, 102, 103):
                if task.value == 101:
                    last = call
                g[:] = P.grad(at, last if task.value == 102 else call)
            elif task.value == 104:
                v = host_view(rv.value, n).copy()
                rec.append(v)
                hv[:] = P.hess_vec(at, v)
            elif task.value == 105:
                f = P.f(at, call)
            out.append(rec)
        free(w)
        return out

    want, got = drive(oracle_backend), drive(hip_backend)
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_[:7] == w_[:7], (which, i, g_[:7], w_[:7])
        for a, b in zip(g_[7:], w_[7:]):
            assert rel_err(a, b) <= TOL, (which, i, rel_err(a, b))


@pytest.mark.parametrize("optname,kw", [
    ("SQN", dict(mem_size=3, bfgs_upd_freq=4)),
    ("oLBFGS", dict(mem_size=4)),
    ("adaQN", dict(mem_size=3, fisher_size=6, bfgs_upd_freq=4, rmsprop_weight=0.9)),
])
def test_export_and_resume_on_P_devices(optname, kw, devices, hip_backend, oracle_backend):
    """Checkpoint of a sharded host caller: export gathers every shard's slices back into the caller's host
    arrays (strided rows of S, Y, F); after the shards are dropped the next call re-imports them."""
    lib = _lib()
    n = 777
    P = NoisyQuadratic(n, seed=5)
    ref = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
    x_ref, x = P.x0(), P.x0()
    compare_traces(run_trace(opt, P, x, 0.05, 41), run_trace(ref, P, x_ref, 0.05, 41), TOL)
    key = C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)
    assert lib.stochqn_hip_devices_active(key) == devices
    assert lib.stochqn_hip_export(key) == 0
    for name in ("s_mem", "y_mem"):
        assert rel_err(getattr(opt.BFGS_mem, name), getattr(ref.BFGS_mem, name)) <= TOL, name
    for name in ("x_sum", "x_avg_prev", "grad_sum_sq", "grad_prev"):
        if hasattr(ref, name) and getattr(ref, name).shape[0] == n:
            assert rel_err(getattr(opt, name), getattr(ref, name)) <= TOL, name
    if hasattr(ref, "Fisher_mem"):
        assert rel_err(opt.Fisher_mem.F, ref.Fisher_mem.F) <= TOL
    lib.stochqn_hip_release(key)
    assert lib.stochqn_hip_devices_active(key) == 0
    P2 = NoisyQuadratic(n, seed=6)
    compare_traces(run_trace(opt, P2, x, 0.05, 40), run_trace(ref, P2, x_ref, 0.05, 40), TOL)
    assert lib.stochqn_hip_devices_active(key) == devices


def test_device_arrays_and_small_problems_stay_on_one_device(hip_backend, oracle_backend):
    """The mode only takes workspaces it can shard: a caller that keeps its arrays in one device's memory
    (torch tensors) and problems below devices_min_n run on the single-device path, unchanged."""
    import torch
    lib = _lib()
    assert lib.stochqn_hip_set_option(b"virtual_devices", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices", 3.0) == 0
    try:
        cfg = [c for c in CONFIGS if c[0] == "sqn_hessvec"][0]
        _, optname, kw, step, calls, pkw = cfg
        P = NoisyQuadratic(1000, seed=7)
        want = run_trace(OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw), P, P.x0(), step, calls)
        # below devices_min_n (default 2^20): host arrays, one device
        opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
        compare_traces(run_trace(opt, P, P.x0(), step, calls), want, TOL)
        assert lib.stochqn_hip_devices_active(C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)) == 0
        # device arrays: one device whatever n
        assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
        opt = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
        x = torch.as_tensor(P.x0(), device="cuda")
        compare_traces(run_trace(opt, P, x, step, calls), want, TOL)
        assert lib.stochqn_hip_devices_active(C.c_void_p(opt.BFGS_mem.s_mem.data_ptr())) == 0
    finally:
        lib.stochqn_hip_release_all()
        lib.stochqn_hip_set_option(b"devices", 0.0)
        lib.stochqn_hip_set_option(b"virtual_devices", 0.0)
        lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


def test_more_devices_than_gpus_without_virtual_devices_falls_back(hip_backend, oracle_backend, capfd):
    """devices = 4 on a one-GPU box without the rehearsal switch: the library says so and runs on what exists."""
    import torch
    lib = _lib()
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices", 4.0) == 0
    try:
        P = NoisyQuadratic(500, seed=3)
        kw = dict(mem_size=3, bfgs_upd_freq=3)
        want = run_trace(OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw), P, P.x0(), 0.1, 30)
        opt = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
        compare_traces(run_trace(opt, P, P.x0(), 0.1, 30), want, TOL)
        assert lib.stochqn_hip_devices_active(C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)) == 0
        assert "only 1 device(s) are visible" in capfd.readouterr().err
    finally:
        lib.stochqn_hip_release_all()
        lib.stochqn_hip_set_option(b"devices", 0.0)
        lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


@pytest.mark.parametrize("profile", ["host_arrays", "library_owned"])
def test_allocation_failures_while_sharding_are_refused_cleanly(profile, hip_backend, oracle_backend, capfd):
    """Every device / pinned allocation of setting a workspace up on 3 shards fails in turn (fault injection): the call
    that needed it returns -1000 / NULL with a message -- no shard is left waiting in a reduction -- and the next
    attempt, with memory available again, runs the oracle's trajectory from the start."""
    lib = _lib()
    assert lib.stochqn_hip_set_option(b"virtual_devices", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices", 3.0) == 0
    n = 1200
    P = NoisyQuadratic(n, seed=2)
    kw = dict(mem_size=3, bfgs_upd_freq=3)
    want = run_trace(OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw), P, P.x0(), 0.1, 20)
    refused = 0
    try:
        for k in range(0, 80):
            lib.stochqn_hip_set_option(b"fail_alloc_after", float(k))
            if profile == "host_arrays":
                opt = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
                x = P.x0()
                try:
                    opt.run_optimizer(x, 0.1)            # section 0: no device work yet
                    opt.update_gradient(P.grad(x, 0))
                    opt.run_optimizer(x, 0.1)            # first step: the group is built here
                    failed = False
                except ValueError:
                    failed = True
                lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
                if failed:
                    refused += 1
                    assert np.array_equal(x, P.x0())     # nothing was touched
                opt.release()
                fresh = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
                compare_traces(run_trace(fresh, P, P.x0(), 0.1, 20), want, TOL)
                fresh.release()
            else:
                w = hip_backend.initialize_SQN(n, 3, 3, 1e-4, 0, 0.0, 1, 1)
                lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
                failed = not bool(w)
                if failed:
                    refused += 1
                else:
                    hip_backend.dealloc_SQN(w)
            if not failed and k > 10:
                break                                    # the injection point lies beyond the last allocation
    finally:
        lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
        lib.stochqn_hip_release_all()
        lib.stochqn_hip_set_option(b"devices", 0.0)
        lib.stochqn_hip_set_option(b"virtual_devices", 0.0)
        lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))
    assert refused >= 10, refused
    assert "could not allocate" in capfd.readouterr().err.lower()


def test_float_abi_on_device_shards(hip_backend):
    """libstochqn_f32.so in the same mode (the group front-end is compiled for both precisions)."""
    import stochqn_amd
    from oracle import oracle
    from test_gpu_parity import F32_TOL
    be32 = stochqn_amd.lib(use_float=True)
    lib32 = stochqn_amd.cdll(use_float=True)
    lib32.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    lib32.stochqn_hip_devices_active.argtypes = [C.c_void_p]
    for name, val in ((b"virtual_devices", 1.0), (b"devices_min_n", 1.0), (b"devices", 3.0)):
        assert lib32.stochqn_hip_set_option(name, val) == 0
    try:
        for cfgname in ("sqn_hessvec", "olbfgs_default", "sqn_graddiff"):      # free-running float trajectories: the tamer configurations
            _, optname, kw, step, calls, pkw = [c for c in CONFIGS if c[0] == cfgname][0]
            P = NoisyQuadratic(2000, seed=7, **pkw)
            want = run_trace(OPTIMIZERS[optname](backend=oracle.bound_f32(), space="host", use_float=True, **kw), P, P.x0().astype(np.float32), step, 40)
            opt = OPTIMIZERS[optname](backend=be32, space="host", use_float=True, **kw)
            got = run_trace(opt, P, P.x0().astype(np.float32), step, 40)
            assert lib32.stochqn_hip_devices_active(C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)) == 3
            compare_traces(got, want, 3 * F32_TOL)
            opt.release()
    finally:
        lib32.stochqn_hip_release_all()
        lib32.stochqn_hip_set_option(b"devices", 0.0)
        lib32.stochqn_hip_set_option(b"virtual_devices", 0.0)
        lib32.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


def _hip_rt():
    import torch
    import os
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip


@pytest.mark.parametrize("which", ["SQN-hessvec", "adaQN-fisher", "oLBFGS"])
def test_device_resident_caller_binds_its_own_shard_vectors(which, devices, hip_backend, oracle_backend):
    """A process that keeps x / grad on the devices: stochqn_hip_devices_layout / _bind / _request.  No vector crosses
    PCIe inside run_*; the requests come back as per-shard device pointers.  Same trajectory as the oracle."""
    import torch
    lib = _lib()
    hip = _hip_rt()
    lib.stochqn_hip_devices_layout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.stochqn_hip_devices_bind.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.stochqn_hip_devices_request.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    n = 10007
    P = NoisyQuadratic(n, seed=4, f_spike_calls=range(30, 36))

    def drive(be, bound):
        x, g, hv = P.x0(), np.zeros(n), np.zeros(n)
        req, rv, task, info = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        if which == "oLBFGS":
            w = be.initialize_oLBFGS(n, 4, 0.0, 0.0, 1e-4, 1, 1)
            run = lambda f: be.run_oLBFGS(0.1, x.ctypes.data, g.ctypes.data, C.byref(req), C.byref(task), w, C.byref(info))
            free = be.dealloc_oLBFGS
        elif which.startswith("SQN"):
            w = be.initialize_SQN(n, 3, 4, 1e-4, 0, 0.0, 1, 1)
            run = lambda f: be.run_SQN(0.1, x.ctypes.data, g.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(rv), C.byref(task), w, C.byref(info))
            free = be.dealloc_SQN
        else:
            w = be.initialize_adaQN(n, 3, 6, 4, 1.01, 1e-4, 1e-4, 0.9, 0, 0.0, 1, 1)
            run = lambda f: be.run_adaQN(0.05, x.ctypes.data, f, g.ctypes.data, C.byref(req), C.byref(task), w, C.byref(info))
            free = be.dealloc_adaQN
        assert bool(w)
        shards = []
        if bound:
            key = C.c_void_p(w.contents.bfgs_memory.contents.s_mem)
            assert lib.stochqn_hip_devices_active(key) == devices
            for p in range(devices):
                dv, off, cnt = C.c_int(), C.c_size_t(), C.c_size_t()
                assert lib.stochqn_hip_devices_layout(key, p, C.byref(dv), C.byref(off), C.byref(cnt)) == 0
                dev = torch.device("cuda", dv.value)
                xs = torch.as_tensor(x[off.value:off.value + cnt.value].copy(), device=dev)
                gs = torch.zeros(cnt.value, dtype=torch.float64, device=dev)
                hs = torch.zeros(cnt.value, dtype=torch.float64, device=dev)
                assert lib.stochqn_hip_devices_bind(key, p, xs.data_ptr(), gs.data_ptr(), hs.data_ptr()) == 0
                shards.append((off.value, cnt.value, xs, gs, hs))
            assert sum(s[1] for s in shards) == n
            x[:] = -7.0                                     # the host x is not the state any more: must stay untouched

        def fetch(ptr_of):                                  # gather a requested vector from the shards' device pointers
            out = np.empty(n)
            for p, (off, cnt, xs, gs, hs) in enumerate(shards):
                rq, rqv = C.c_void_p(), C.c_void_p()
                assert lib.stochqn_hip_devices_request(key, p, C.byref(rq), C.byref(rqv)) == 0
                src = ptr_of(rq.value, rqv.value)
                if src == xs.data_ptr():
                    out[off:off + cnt] = xs.cpu().numpy()
                else:
                    assert hip.hipMemcpy(out[off:off + cnt].ctypes.data, src, 8 * cnt, 2) == 0
            return out

        def scatter(vec, idx):
            for off, cnt, *ts in shards:
                ts[idx].copy_(torch.as_tensor(vec[off:off + cnt]))

        out, f, last = [], 0.0, 0
        for call in range(50):
            rc = run(f)
            assert rc in (0, 1), rc
            if bound:
                at = fetch(lambda a, b: a)
                xnow = np.concatenate([s[2].cpu().numpy() for s in shards])
            else:
                at, xnow = host_view(req.value, n).copy(), x.copy()
            rec = [rc, task.value, info.value, w.contents.niter, w.contents.section,
                   w.contents.bfgs_memory.contents.mem_used, w.contents.bfgs_memory.contents.mem_st_ix, xnow, at]
            if task.value in (101, 102, 103):
                if task.value == 101:
                    last = call
                gv = P.grad(at, last if task.value == 102 else call)
                if bound:
                    scatter(gv, 1)
                else:
                    g[:] = gv
            elif task.value == 104:
                v = fetch(lambda a, b: b) if bound else host_view(rv.value, n).copy()
                rec.append(v)
                if bound:
                    scatter(P.hess_vec(at, v), 2)
                else:
                    hv[:] = P.hess_vec(at, v)
            elif task.value == 105:
                f = P.f(at, call)
            out.append(rec)
        if bound:
            assert np.all(x == -7.0)
        free(w)
        return out

    want, got = drive(oracle_backend, False), drive(hip_backend, True)
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_[:7] == w_[:7], (which, i, g_[:7], w_[:7])
        for a, b in zip(g_[7:], w_[7:]):
            assert rel_err(a, b) <= TOL, (which, i, rel_err(a, b))


def test_foreach_runs_the_callers_shard_work_under_the_groups_reducer(devices, hip_backend):
    """stochqn_hip_devices_foreach: the callback runs on every shard's own thread, with the shard's device current and
    its reducer bound -- so an isolated entry point called from it reduces over ALL shards: the sharded
    Fisher product t = F s, y = F't / fu equals the oracle's product on the whole matrix."""
    import threading
    import torch
    from oracle import oracle
    lib = _lib()
    SHARD_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t)
    lib.stochqn_hip_devices_foreach.argtypes = [C.c_void_p, SHARD_FN, C.c_void_p]
    lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n, fu = 50021, 9
    rng = np.random.default_rng(1)
    F, s = rng.standard_normal((fu, n)), rng.standard_normal(n)
    t_w, y_w = oracle.fisher_product(F.reshape(-1).copy(), fu, s)
    w = hip_backend.initialize_SQN(n, 3, 4, 0.0, 0, 0.0, 1, 1)
    key = C.c_void_p(w.contents.bfgs_memory.contents.s_mem)
    seen, results, errors = {}, {}, []

    def work(user, shard, device, off, cnt):
        try:
            seen[shard] = (device, off, cnt, threading.get_ident())
            dev = torch.device("cuda", device)
            Fp = torch.as_tensor(np.ascontiguousarray(F[:, off:off + cnt]).reshape(-1), device=dev)
            sp = torch.as_tensor(s[off:off + cnt].copy(), device=dev)
            yp = torch.zeros(cnt, dtype=torch.float64, device=dev)
            t = np.zeros(fu)
            rc = lib.stochqn_hip_fisher_product(Fp.data_ptr(), fu, cnt, sp.data_ptr(), t.ctypes.data, yp.data_ptr())
            lib.stochqn_hip_release(C.c_void_p(Fp.data_ptr()))
            results[shard] = (rc, t, yp.cpu().numpy(), off, cnt)
        except Exception as e:                               # pragma: no cover
            errors.append(repr(e))

    keep = SHARD_FN(work)
    assert lib.stochqn_hip_devices_foreach(key, keep, None) == 0
    hip_backend.dealloc_SQN(w)
    assert not errors, errors
    assert sorted(seen) == list(range(devices))
    assert len({v[3] for v in seen.values()}) == devices            # one host thread per shard
    assert sum(v[2] for v in seen.values()) == n
    y = np.empty(n)
    for rc, t, yp, off, cnt in results.values():
        assert rc == 0
        assert rel_err(t, t_w) <= TOL                                # every shard holds the GLOBAL t
        y[off:off + cnt] = yp
    assert rel_err(y, y_w) <= TOL


def test_bench_in_process_mode_is_shard_invariant(tmp_path):
    """bench.py --in-process (one process, device-resident caller of the multi-device mode, rehearsed on one GPU): the
    same n_total cut into 2 and into 3 shards must give the same x (counter-based inputs, identical decisions)."""
    import json
    from test_gpu_parity import _bench
    common = ["--in-process", "--virtual-devices", "--mem", "4", "--upd-freq", "3", "--steps", "9", "--warmup", "2", "--no-cpu-baseline"]
    outs = {}
    for P, per in ((2, 1_500_000), (3, 1_000_000)):
        r = _bench(["--gpus", str(P), "--vars-per-gpu", str(per), "--dump-x", str(tmp_path / ("x%d" % P))] + common)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == P and d["device_shards"] == P and d["config"]["hess_vec_requests"] >= 2
        assert d["config"]["rejected_steps"] == 0 and d["config"]["f_end"] < d["config"]["f_start"]
        outs[P] = (d, np.load(str(tmp_path / ("x%d.0.npy" % P))))
    assert outs[2][1].shape == outs[3][1].shape == (3_000_000,)
    assert rel_err(outs[2][1], outs[3][1]) <= TOL
    assert outs[2][0]["config"]["calls"] == outs[3][0]["config"]["calls"]
    # the same process model with a HOST caller (numpy arrays, each shard moving its slice over its own link): reported per shard
    for P in (2, 3):
        h = outs[P][0]["host_caller"]
        assert "error" not in h, h
        assert h["arrays_pinned_by_the_caller"] == 5 and h["ordinary_step_ms"] > 0 and len(h["per_step_ms"]) == 3
        assert h["bytes_up_per_shard"] == 2 * 8 * 4_000_000 and h["link_GBps_per_shard_up"] > 0 and h["link_GBps_per_shard_down"] > 0
        assert h["pcie_probe_per_device_alone"]["cuda:0"]["h2d_GBps"] > 5


def test_c5_shard_size_through_the_single_process_mode():
    """BASELINE config 5's per-GPU shard (n = 1.25e8, m = 20: 40 GB of S and Y per shard) through the single-process
    multi-device mode: two such shards (n_total = 2.5e8) rehearsed on the one GPU -- initialize_SQN on shards, the ring
    filled by the run itself, the 32-row Hessian product reduced over the shards, full-size kernels."""
    import json
    from test_gpu_parity import _bench
    r = _bench(["--gpus", "2", "--in-process", "--virtual-devices", "--config", "c5", "--steps", "10", "--warmup", "2", "--no-host-caller"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["device_shards"] == 2 and "n_total=2.5e+08" in d["config"]["workload"]
    assert d["config"]["hess_vec_requests"] == 1 and d["config"]["rejected_steps"] == 0 and d["config"]["rejected_pairs"] == 0
    assert d["config"]["f_end"] < d["config"]["f_start"]
    assert d["roofline"]["alg_bytes_per_launch"] in (21 * 125_000_000 * 8, 22 * 125_000_000 * 8)      # a pass of the three-pass form


def test_shard_groups_do_not_leak_device_memory_or_threads(hip_backend):
    """Sixty sharded workspaces -- host-array ones dropped by release / a new optimiser at the same address, library-owned
    ones by dealloc_* -- must leave the device's free memory and the process's thread count where they were (every group
    owns P worker threads, P device contexts and the slices of every array)."""
    import threading
    import torch
    lib = _lib()
    for name, val in ((b"virtual_devices", 1.0), (b"devices_min_n", 1.0), (b"devices", 3.0)):
        assert lib.stochqn_hip_set_option(name, val) == 0
    n = 40_000
    P = NoisyQuadratic(n, seed=3)

    def cycle():
        for optname, kw in (("SQN", dict(mem_size=6, bfgs_upd_freq=3)), ("adaQN", dict(mem_size=5, fisher_size=9, bfgs_upd_freq=3, max_incr=None)),
                            ("oLBFGS", dict(mem_size=6))):
            opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
            run_trace(opt, P, P.x0(), 0.05, 10)
            opt.release()
        w = hip_backend.initialize_SQN(n, 5, 3, 1e-4, 0, 0.0, 1, 1)
        assert bool(w)
        hip_backend.dealloc_SQN(w)

    def threads():
        import os
        return len(os.listdir("/proc/self/task"))

    try:
        cycle()
        lib.stochqn_hip_release_all()
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        t0 = threads()
        for _ in range(15):
            cycle()
        lib.stochqn_hip_release_all()
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free0 - free1 < 2 * 6 * n * 8, "device memory shrank by %d bytes over 60 sharded workspaces" % (free0 - free1)
        assert threads() <= t0 + 2, (t0, threads())
    finally:
        lib.stochqn_hip_release_all()
        lib.stochqn_hip_set_option(b"devices", 0.0)
        lib.stochqn_hip_set_option(b"virtual_devices", 0.0)
        lib.stochqn_hip_set_option(b"devices_min_n", float(1 << 20))


def test_process_exit_with_live_shard_groups_is_clean(tmp_path):
    """An R session (or a script) that just quits: sharded workspaces never released nor deallocated, their worker threads
    idle on their condition variables.  The process must still end normally."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "quit.py"
    script.write_text('''
import ctypes as C, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import stochqn_amd
from harness import NoisyQuadratic, run_trace, OPTIMIZERS
lib = stochqn_amd.cdll(); lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
for k, v in ((b"virtual_devices", 1.0), (b"devices_min_n", 1.0), (b"devices", 3.0)):
    assert lib.stochqn_hip_set_option(k, v) == 0
P = NoisyQuadratic(5000, seed=1)
opt = OPTIMIZERS["SQN"](backend=stochqn_amd.lib(), space="host", mem_size=4, bfgs_upd_freq=3)
run_trace(opt, P, P.x0(), 0.1, 20)
opt.release = lambda: None                                  # no release from __del__ either
w = stochqn_amd.lib().initialize_SQN(5000, 3, 3, 0.0, 0, 0.0, 1, 1)
assert bool(w)
print("leaving", flush=True)
''' % (root, os.path.join(root, "tests")))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "leaving" in out.stdout
