"""GPU parity tests proper: the HIP library through its C ABI against the CPU oracle and against
the reference's own known answers.  Tolerance: 1e-10 relative (fp64), the figure BASELINE.json's
north_star states; everything discrete (task / info / section / counters / which array *req
aliases) must match exactly."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from harness import OPTIMIZERS, NoisyQuadratic, compare_traces, rel_err, run_lockstep, run_trace, to_dev, to_np
from test_oracle_known_answers import GOLD, check_known_answer, run_c_rosen, host_view

pytestmark = pytest.mark.gpu
TOL = 1e-10


FORMS = {"threepass": 1.0, "sweeps": 0.0}      # name -> option "threepass"
# The chain of sweeps is element-wise (no tiles, no row split): sizes in the middle of a grid add nothing to it that the ends
# (1, one past a wave, one past a pack, the large odd sizes) do not show; its full launch shape runs at n = 1,000,003
# (test_lockstep_parity_full_grids).  The default form runs every size.
SWEEPS_SKIP_N = {2, 7, 63, 64, 127, 129, 1000, 70001}


def set_form(lib, name):
    """Select the implementation of the two-loop recursion: the three-pass form (default: S twice, Y once) or the
    reference's chain of dependent sweeps.  The isolated entry points only use the cached form when the caller
    vouches for the arrays (raw_reuse_cache); the tests release their contexts after every call."""
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    th = FORMS[name]
    assert lib.stochqn_hip_set_option(b"threepass", th) == 0
    assert lib.stochqn_hip_set_option(b"raw_reuse_cache", th) == 0


def reset_form(lib):
    set_form(lib, "threepass")
    lib.stochqn_hip_set_option(b"raw_reuse_cache", 0.0)


@pytest.fixture(params=["threepass", "sweeps"])
def form(request, hip_backend):
    """Run a test once per implementation of the two-loop: the default three-pass form and the chain of sweeps."""
    import stochqn_amd
    if request.param == "sweeps" and getattr(request.node, "callspec", None) is not None \
            and request.node.callspec.params.get("n") in SWEEPS_SKIP_N:
        pytest.skip("the fallback form runs the ends of each size grid, the default form all of it")
    lib = stochqn_amd.cdll()
    set_form(lib, request.param)
    yield request.param
    reset_form(lib)


def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


# ---------------------------------------------------------------------------------------------
# reference known answers through the product
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("space", ["host", "device"])
@pytest.mark.parametrize("case", ["oLBFGS_rosen2d", "SQN_rosen2d", "adaQN_rosen2d"])
def test_known_answers(case, space, form, hip_backend):
    # held to the ORACLE'S OWN pin (rtol 1e-11 / 1e-13 / 1e-12 per case; rounds 1 - 4 allowed 1e3 x that).  Measured on the MI355X in
    # round 5, host and device callers alike: oLBFGS 3.9e-13 (three-pass form; 0 in the sweeps), SQN 1.4e-16, adaQN 1.7e-15
    # (gpurun_out/known_answers_errors.txt of the run, quoted in DESIGN.md section 6)
    check_known_answer(case, hip_backend, space=space, tol_scale=1.0)
    try:                                                      # the measured error, for the record of the run (gpurun_out/ is scratch)
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "known_answers_errors.txt"), "a") as f:
            f.write("%s %s %s %.3e\n" % (case, space, form, check_known_answer.measured[(case, space)]))
    except OSError:
        pass


def test_c_rosen_protocol_host_caller(form, hip_backend):
    """Library-owned workspace (arrays in HBM) driven by a host caller exactly like c_rosen.c."""
    k = GOLD["c_rosen"]
    out = run_c_rosen(hip_backend, np.array(k["x0"]), host_view)
    for key in ("f_initial", "f_it10", "f_it200", "f_final"):
        assert out[key].strip() == k[key], (key, out[key])
    assert out["x_final"] == k["x_final"]


# ---------------------------------------------------------------------------------------------
# per-call trace parity on noisy quadratics
# ---------------------------------------------------------------------------------------------
CONFIGS = [
    # name, optimizer, kwargs, step, calls, problem kwargs
    ("olbfgs_default", "oLBFGS", dict(mem_size=4), 0.1, 60, {}),
    ("olbfgs_nocurv_hess_init", "oLBFGS", dict(mem_size=3, min_curvature=None, hess_init=0.5, y_reg=1e-3), 0.05, 50, {}),
    ("olbfgs_nonan_check", "oLBFGS", dict(mem_size=5, check_nan=False, min_curvature=None), 0.1, 50, {}),
    ("olbfgs_nan_grad", "oLBFGS", dict(mem_size=4), 0.1, 60, dict(nan_calls=(21, 22))),
    ("olbfgs_reject_all", "oLBFGS", dict(mem_size=2, min_curvature=10.0), 0.1, 30, {}),
    ("sqn_hessvec", "SQN", dict(mem_size=3, bfgs_upd_freq=4), 0.1, 80, {}),
    ("sqn_hessvec_L1", "SQN", dict(mem_size=3, bfgs_upd_freq=1, min_curvature=None), 0.1, 30, {}),
    ("sqn_graddiff", "SQN", dict(mem_size=3, bfgs_upd_freq=5, use_grad_diff=True, y_reg=1e-2), 0.1, 90, {}),
    ("sqn_reject", "SQN", dict(mem_size=2, bfgs_upd_freq=3, min_curvature=5.0), 0.1, 60, {}),
    ("sqn_nan", "SQN", dict(mem_size=3, bfgs_upd_freq=4), 0.1, 60, dict(nan_calls=(30,))),
    ("sqn_nonan_check", "SQN", dict(mem_size=3, bfgs_upd_freq=4, check_nan=False), 0.1, 60, {}),
    ("adaqn_fisher_rms", "adaQN", dict(mem_size=3, fisher_size=7, bfgs_upd_freq=4, rmsprop_weight=0.9), 0.05, 90, {}),
    ("adaqn_fisher_adagrad_nomaxincr", "adaQN", dict(mem_size=3, fisher_size=5, bfgs_upd_freq=3, max_incr=None), 0.05, 70, {}),
    ("adaqn_graddiff", "adaQN", dict(mem_size=3, bfgs_upd_freq=4, use_grad_diff=True, rmsprop_weight=0.5), 0.05, 90, {}),
    ("adaqn_graddiff_nomaxincr", "adaQN", dict(mem_size=2, bfgs_upd_freq=3, use_grad_diff=True, max_incr=None), 0.05, 60, {}),
    ("adaqn_func_increased", "adaQN", dict(mem_size=3, fisher_size=6, bfgs_upd_freq=4), 0.05, 90, dict(f_spike_calls=range(40, 60))),
    ("adaqn_nan", "adaQN", dict(mem_size=3, fisher_size=6, bfgs_upd_freq=4), 0.05, 70, dict(nan_calls=(33,))),
    ("adaqn_nonan_check", "adaQN", dict(mem_size=3, fisher_size=6, bfgs_upd_freq=4, check_nan=False), 0.05, 70, {}),
    # a Fisher ring larger than any fixed-size scalar buffer in the library (and never full in this run)
    ("adaqn_fisher500", "adaQN", dict(mem_size=3, fisher_size=500, bfgs_upd_freq=4, max_incr=None), 0.05, 40, {}),
    # rings of 25 .. 48 pairs: the three-pass form still applies; beyond 48 the step runs as the sweeps (sqn_ring50).
    # (The adaQN ring
    # configs stop after ~30 calls: later the iterates jitter around the optimum, s = x_avg - x_avg_prev
    # has mixed signs and F s cancels to ~1e-6 of its terms, so ANY two summation orders differ by
    # ~1e-10 in y -- a property of the instance, not of the kernel.)
    ("sqn_ring30", "SQN", dict(mem_size=30, bfgs_upd_freq=1, min_curvature=None), 0.05, 80, {}),
    ("sqn_ring48", "SQN", dict(mem_size=48, bfgs_upd_freq=1, min_curvature=None), 0.05, 110, {}),
    ("sqn_ring50", "SQN", dict(mem_size=50, bfgs_upd_freq=1, min_curvature=None), 0.05, 110, {}),
    ("olbfgs_ring26", "oLBFGS", dict(mem_size=26, min_curvature=None), 0.05, 70, {}),
    ("adaqn_ring25", "adaQN", dict(mem_size=25, fisher_size=8, bfgs_upd_freq=1, max_incr=None, min_curvature=None), 0.002, 31, {}),
    # a full 20-pair ring (the BASELINE shape) at test size
    ("sqn_ring20", "SQN", dict(mem_size=20, bfgs_upd_freq=1, min_curvature=None), 0.05, 60, {}),
    ("adaqn_ring20", "adaQN", dict(mem_size=20, fisher_size=16, bfgs_upd_freq=1, max_incr=None, min_curvature=None, rmsprop_weight=0.9), 0.002, 30, {}),
]


# Free-running trajectories that amplify last-bit differences: the bar is what TWO LEGAL EVALUATIONS OF THE REFERENCE'S OWN
# ARITHMETIC keep between them, measured on the CPU with the kernels out of the question (round 6):
# profiles/r06_oracle_vs_oracle_sensitivity.json is the oracle against itself with nothing changed but the order in which its
# dot products are summed (8 / 4 / 1 interleaved partial sums, OpenBLAS: every one an order a BLAS may choose), same
# configurations, same sizes; profiles/r06_free_run_device_vs_oracle.json is the library on an MI355X against the oracle.
#   adaqn_fisher_adagrad_nomaxincr  oracle vs oracle up to 1.0e-7 (n = 70001; 7.8e-9 at n = 4097), device vs oracle 5.9e-8 / 4.3e-9
#   adaqn_ring20                    oracle vs oracle up to 2.1e-10 (n = 4097), device vs oracle 2.4e-10: held to 2e-9 (was 1e-7)
#   adaqn_ring25                    oracle vs oracle 1.6e-12, device vs oracle 9.2e-13: held to 1e-10 like everything else (was 1e-7)
# -- the device sits where any other summation order sits, configuration by configuration and size by size; the per-call bar
# of 1e-10 is enforced for all of them by the lock-step test below.  tests/test_oracle_sensitivity.py keeps this table honest.
FREE_RUN_TOL = {"adaqn_fisher_adagrad_nomaxincr": 1e-7, "adaqn_ring20": 2e-9}


def both_traces(cfg, n, space, hip_backend, oracle_backend):
    name, optname, kw, step, calls, pkw = cfg
    P = NoisyQuadratic(n, seed=7, **pkw)
    ref_opt = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    want = run_trace(ref_opt, P, P.x0(), step, calls)
    opt = OPTIMIZERS[optname](backend=hip_backend, space=space, **kw)
    x = P.x0()
    if space == "device":
        x = torch_cuda().as_tensor(x, device="cuda")
    got = run_trace(opt, P, x, step, calls)
    return got, want


@pytest.mark.parametrize("n", [1, 2, 7, 64, 1000, 4097])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_trace_parity_device_arrays(cfg, n, form, hip_backend, oracle_backend):
    got, want = both_traces(cfg, n, "device", hip_backend, oracle_backend)
    compare_traces(got, want, FREE_RUN_TOL.get(cfg[0], TOL))


@pytest.mark.parametrize("n", [3, 64, 1000, 4097, 70001])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_lockstep_parity(cfg, n, form, hip_backend, oracle_backend):
    """Identical state and inputs into the oracle and the HIP library on every call; every output
    array, every state array and every scalar compared after every call."""
    import stochqn_amd
    name, optname, kw, step, calls, pkw = cfg
    if name == "adaqn_fisher500" and n > 5000:
        pytest.skip("500 Fisher rows x 70001 copied back and forth every call: 12 s for no new code path")
    if kw.get("mem_size", 0) >= 25 and n > 5000:
        pytest.skip("rings of 25-50 pairs x 70001 copied back and forth every call: the same kernels as at n = 4097")
    P = NoisyQuadratic(n, seed=11, **pkw)
    ref = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
    x_ref = P.x0()
    x_dev = torch_cuda().as_tensor(P.x0(), device="cuda")
    lib = stochqn_amd.cdll()
    inval = lambda o: lib.stochqn_hip_invalidate(C.c_void_p(o._sp.ptr(o.BFGS_mem.s_mem)))
    run_lockstep(ref, opt, P, x_ref, x_dev, step, min(calls, 60), TOL, on_sync=inval)


@pytest.mark.parametrize("cfgname", ["sqn_ring20", "adaqn_ring20", "olbfgs_default"])
def test_lockstep_parity_full_grids(cfgname, form, hip_backend, oracle_backend):
    """The same lock-step comparison at a size where every kernel runs its full launch shape: one
    workgroup per CU in the sweeps, the whole-rounds grid of the row-split pass 1, full grids in passes 2 and 3,
    and -- n odd -- every other ring row off the 16-byte grid.  (The sweeps take the two configurations whose
    operators differ, SQN's and oLBFGS's; adaQN's chain is SQN's plus the diagonal, covered at the smaller sizes.)"""
    import stochqn_amd
    if form == "sweeps" and cfgname == "adaqn_ring20":
        pytest.skip("the sweeps run sqn_ring20 and olbfgs_default at this size")
    name, optname, kw, step, calls, pkw = {c[0]: c for c in CONFIGS}[cfgname]
    n = 1_000_003
    P = NoisyQuadratic(n, seed=11, **pkw)
    ref = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
    x_ref = P.x0()
    x_dev = torch_cuda().as_tensor(P.x0(), device="cuda")
    lib = stochqn_amd.cdll()
    inval = lambda o: lib.stochqn_hip_invalidate(C.c_void_p(o._sp.ptr(o.BFGS_mem.s_mem)))
    run_lockstep(ref, opt, P, x_ref, x_dev, step, min(calls, 20), TOL, on_sync=inval)



@pytest.mark.parametrize("n", [2, 65, 1000])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_trace_parity_host_arrays(cfg, n, form, hip_backend, oracle_backend):
    """Profile B of SURVEY.md 8b: every array in host memory, structs rebuilt per call."""
    got, want = both_traces(cfg, n, "host", hip_backend, oracle_backend)
    compare_traces(got, want, FREE_RUN_TOL.get(cfg[0], TOL))


def test_golden_traces(form, hip_backend):
    """Committed regression vectors (tests/golden/traces.json, made by tests/golden/make_traces.py)."""
    path = os.path.join(os.path.dirname(__file__), "golden", "traces.json")
    gold = json.load(open(path))
    by_name = {c[0]: c for c in CONFIGS}
    for name, entry in gold["traces"].items():
        cfg = by_name[name]
        P = NoisyQuadratic(entry["n"], seed=7, **cfg[5])
        opt = OPTIMIZERS[cfg[1]](backend=hip_backend, space="device", **cfg[2])
        x = torch_cuda().as_tensor(P.x0(), device="cuda")
        got = run_trace(opt, P, x, cfg[3], cfg[4])
        compare_traces(got, entry["trace"], TOL)


# ---------------------------------------------------------------------------------------------
# isolated two-loop recursion (reference src/stochqn.c:663-708)
# ---------------------------------------------------------------------------------------------
def make_pairs(rng, n, m):
    d = 0.5 + rng.random(n)
    S = 1e-3 * (rng.random((m, n)) - 0.5)
    Y = S * d
    return S.reshape(-1).copy(), Y.reshape(-1).copy()


def hip_two_loop(lib, g, H0, h0, Y, S, n, m, used, st):
    rho = np.zeros(m)
    alpha = np.zeros(m)
    lib.stochqn_hip_two_loop.restype = C.c_int
    lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    rc = lib.stochqn_hip_two_loop(g.data_ptr(), n, None if H0 is None else H0.data_ptr(), h0, Y.data_ptr(),
                                  S.data_ptr(), m, used, st, rho.ctypes.data, alpha.ctypes.data)
    assert rc == 0
    return rho, alpha


TWO_LOOP_SHAPES = [(1, 1, 0), (5, 5, 3), (5, 2, 0), (5, 3, 4), (20, 20, 7), (20, 1, 19)]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4096, 1000003])
@pytest.mark.parametrize("m,used,st", TWO_LOOP_SHAPES)
@pytest.mark.parametrize("mode", ["gamma", "h0"])
def test_two_loop_matches_oracle(n, m, used, st, mode, form, hip_backend):
    """Scalar H0 (gamma from the newest pair, or h0 > 0): both forms of the recursion."""
    check_two_loop(n, m, used, st, mode)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4096, 1000003])
@pytest.mark.parametrize("m,used,st", TWO_LOOP_SHAPES)
def test_two_loop_with_a_given_diagonal_matches_oracle(n, m, used, st, form, hip_backend):
    """A caller-supplied diagonal H0: the three-pass form scales q0 by it in pass 2 (k_qdot, MODE 1)."""
    check_two_loop(n, m, used, st, "H0")


def check_two_loop(n, m, used, st, mode):
    import stochqn_amd
    from oracle import oracle
    torch = torch_cuda()
    rng = np.random.default_rng(n * 131 + m * 7 + used)
    S, Y = make_pairs(rng, n, m)
    g = rng.random(n) - 0.5
    H0 = (0.5 + rng.random(n)) if mode == "H0" else None
    h0 = 0.37 if mode == "h0" else 0.0

    want = g.copy()
    rho_w, alpha_w = oracle.two_loop(want, H0, h0, Y, S, m, used, st)

    dS, dY, dg = (torch.as_tensor(a, device="cuda") for a in (S, Y, g))
    dH0 = None if H0 is None else torch.as_tensor(H0, device="cuda")
    rho, alpha = hip_two_loop(stochqn_amd.cdll(), dg, dH0, h0, dY, dS, n, m, used, st)
    stochqn_amd.cdll().stochqn_hip_release(C.c_void_p(dS.data_ptr()))

    assert rel_err(dg.cpu().numpy(), want) <= TOL
    assert np.allclose(rho[:used], rho_w[:used], rtol=TOL, atol=0)
    # buffer_alpha is scratch in the reference; entries that are pure cancellation noise (n = 1: all pairs
    # collinear, every alpha but the newest is 0 in exact arithmetic) are compared against the largest one
    assert np.allclose(alpha[:used], alpha_w[:used], rtol=1e-9, atol=1e-13 * np.abs(alpha_w[:used]).max())


@pytest.mark.parametrize("n,m,grid_cap", [(200_001, 20, 4), (70_002, 33, 2), (3_000_001, 20, 0), (1000, 5, 1)])
@pytest.mark.parametrize("mode", ["gamma", "H0"])
def test_clock_phased_stores_change_nothing_but_the_moment(n, m, grid_cap, mode, hip_backend):
    """Passes 2 and 3 of the three-pass form park their results in LDS and store them when the chip-wide clock enters a new
    period (`phase_ticks`, kernels.hip: Parked).  Whatever the period -- the default, one so short that every few iterations
    flush, one so long that only full slots do -- and with the phases off (every pack stored at once), the direction, rho and
    alpha are the same BITS: the same values reach the same addresses, only later.  A small grid_cap gives every lane hundreds
    of packs (its 32 slots fill many times over) at a size the test can afford; the result is also held to the oracle."""
    import stochqn_amd
    from oracle import oracle
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    rng = np.random.default_rng(n + m)
    S, Y = make_pairs(rng, n, m)
    g = rng.random(n) - 0.5
    H0 = (0.5 + rng.random(n)) if mode == "H0" else None
    used, st = m, 3 % m
    want = g.copy()
    oracle.two_loop(want, H0, 0.0, Y, S, m, used, st)
    dS, dY = (torch.as_tensor(a, device="cuda") for a in (S, Y))
    dH0 = None if H0 is None else torch.as_tensor(H0, device="cuda")
    got = {}
    try:
        assert lib.stochqn_hip_set_option(b"raw_reuse_cache", 1.0) == 0        # the isolated entry in the three-pass form
        assert lib.stochqn_hip_set_option(b"grid_cap", float(grid_cap)) == 0
        for ticks in (8000, 0, 3, 50, 100000000):
            assert lib.stochqn_hip_set_option(b"phase_ticks", float(ticks)) == 0
            dg = torch.as_tensor(g, device="cuda")
            rho, alpha = hip_two_loop(lib, dg, dH0, 0.0, dY, dS, n, m, used, st)
            got[ticks] = (dg.cpu().numpy(), rho, alpha)
            lib.stochqn_hip_release(C.c_void_p(dS.data_ptr()))
    finally:
        lib.stochqn_hip_set_option(b"phase_ticks", 8000.0)
        lib.stochqn_hip_set_option(b"grid_cap", 0.0)
        lib.stochqn_hip_set_option(b"raw_reuse_cache", 0.0)
        lib.stochqn_hip_release(C.c_void_p(dS.data_ptr()))
    assert rel_err(got[8000][0], want) <= TOL
    for ticks in (0, 3, 50, 100000000):
        for a, b in zip(got[ticks], got[8000]):
            assert np.array_equal(a, b), "phase_ticks = %d differs from the default" % ticks


@pytest.mark.parametrize("kind,use_float", [("SQN", False), ("adaQN", False), ("adaQN", True), ("oLBFGS", True)])
def test_clock_phased_stores_leave_whole_runs_bit_identical(kind, use_float, hip_backend, hip_backend_f32):
    """The same over whole optimiser runs, device-resident, in both builds: adaQN's pass 2 parks four vectors per pack (r0, G, H0
    and the Fisher row), its pairs come out of the Fisher product, whose second pass parks y.  Every x, every request, every
    counter of 40 calls with the phases on, off, and at a period of a few hundred nanoseconds: identical to the last bit.
    (grid_cap = 2: two workgroups walk the whole vector, so every lane fills its slots many times at n = 300,001.)"""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll(use_float)
    be = hip_backend_f32 if use_float else hip_backend
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    n = 300_001
    P = NoisyQuadratic(n, seed=21)
    kw = {"SQN": dict(mem_size=9, bfgs_upd_freq=3), "oLBFGS": dict(mem_size=9),
          "adaQN": dict(mem_size=9, fisher_size=11, bfgs_upd_freq=3, max_incr=1.01, rmsprop_weight=0.9)}[kind]
    dt = np.float32 if use_float else np.float64
    runs = {}
    try:
        assert lib.stochqn_hip_set_option(b"grid_cap", 2.0) == 0
        for ticks in (8000, 0, 40):
            assert lib.stochqn_hip_set_option(b"phase_ticks", float(ticks)) == 0
            x = torch.as_tensor(P.x0().astype(dt), device="cuda")
            runs[ticks] = run_trace(OPTIMIZERS[kind](backend=be, space="device", device="cuda", use_float=use_float, **kw), P, x, 0.05, 40)
            lib.stochqn_hip_release_all()
    finally:
        lib.stochqn_hip_set_option(b"phase_ticks", 8000.0)
        lib.stochqn_hip_set_option(b"grid_cap", 0.0)
        lib.stochqn_hip_release_all()
    assert max(t["mem_used"] for t in runs[8000]) >= 5, "the ring never filled far enough for the three-pass form to matter"
    for ticks in (0, 40):
        for i, (a, b) in enumerate(zip(runs[8000], runs[ticks])):
            assert a.keys() == b.keys()
            for k in a:
                same = np.array_equal(a[k], b[k], equal_nan=True) if isinstance(a[k], np.ndarray) else (a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]))
                assert same, "phase_ticks = %d, call %d: %s differs" % (ticks, i, k)


def test_two_loop_host_pointers(hip_backend):
    """Same entry point fed with plain numpy memory (staged over PCIe behind the ABI)."""
    import stochqn_amd
    from oracle import oracle
    lib = stochqn_amd.cdll()
    n, m, used, st = 1001, 4, 4, 2
    rng = np.random.default_rng(5)
    S, Y = make_pairs(rng, n, m)
    g = rng.random(n) - 0.5
    want = g.copy()
    oracle.two_loop(want, None, 0.0, Y, S, m, used, st)
    rho = np.zeros(m)
    alpha = np.zeros(m)
    lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    assert lib.stochqn_hip_two_loop(g.ctypes.data, n, None, 0.0, Y.ctypes.data, S.ctypes.data, m, used, st,
                                    rho.ctypes.data, alpha.ctypes.data) == 0
    lib.stochqn_hip_release(C.c_void_p(S.ctypes.data))
    assert rel_err(g, want) <= TOL


# ---------------------------------------------------------------------------------------------
# take_step on its own (reference src/stochqn.c:802-840): the entry that reaches adaQN's step (pass 2 in its
# adaQN mode, or the sweeps with the diagonal H0) with a state of the caller's choice
# ---------------------------------------------------------------------------------------------
def take_step_args(lib):
    from stochqn_amd import _abi
    lib.stochqn_hip_take_step.restype = C.c_int
    lib.stochqn_hip_take_step.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_abi.bfgs_mem), C.c_double,
                                          C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int)]
    return _abi


def oracle_take_step(step, x, g, S, Y, m, used, st_ix, w, H0, h0, G, eps, check_nan):
    """In place on numpy arrays; returns (info, mem_used, rho, alpha)."""
    from oracle import oracle
    from stochqn_amd import _abi
    n = x.shape[0]
    rho, alpha = np.zeros(m), np.zeros(m)
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho.ctypes.data, alpha.ctypes.data, None, None, m, used, st_ix, 1, 0.0, 0.0)
    info = C.c_int(200)
    oracle.cdll().oracle_take_step(step, n, x.ctypes.data, g.ctypes.data, C.byref(b), w, None if H0 is None else H0.ctypes.data,
                                   h0, None if G is None else G.ctypes.data, eps, check_nan, C.byref(info))
    return info.value, b.mem_used, rho, alpha


def hip_take_step(lib, step, x, g, S, Y, m, used, st_ix, w, H0, h0, G, eps, check_nan):
    """On torch device tensors; returns (info, mem_used, rho, alpha)."""
    _abi = take_step_args(lib)
    rho, alpha = np.zeros(m), np.zeros(m)
    b = _abi.bfgs_mem(S.data_ptr(), Y.data_ptr(), rho.ctypes.data, alpha.ctypes.data, None, None, m, used, st_ix, 1, 0.0, 0.0)
    info = C.c_int(200)
    rc = lib.stochqn_hip_take_step(step, x.shape[0], x.data_ptr(), g.data_ptr(), C.byref(b), w, None if H0 is None else H0.data_ptr(),
                                   h0, None if G is None else G.data_ptr(), eps, check_nan, C.byref(info))
    assert rc == 0
    return info.value, b.mem_used, rho, alpha


@pytest.mark.parametrize("n", [1, 65, 4096, 1000003])
@pytest.mark.parametrize("m,used,st_ix", [(5, 5, 3), (5, 2, 2), (20, 20, 7), (20, 20, 0), (3, 0, 0), (1, 1, 0)])
@pytest.mark.parametrize("mode", ["rmsprop", "adagrad", "gamma", "h0"])
def test_take_step_matches_oracle(n, m, used, st_ix, mode, form, hip_backend):
    """Direction, x, G, H0, rho, alpha and the verdict of one isolated step, in the three-pass form (k_qdot in its adaQN
    mode for the two diagonal modes) and as the chain of dependent sweeps."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    rng = np.random.default_rng(n * 17 + m * 5 + used + len(mode))
    S, Y = make_pairs(rng, n, m)
    g = rng.random(n) - 0.5
    x = 1.0 + rng.random(n)
    diag = mode in ("rmsprop", "adagrad")
    G = (0.1 + rng.random(n)) if diag else None
    H0 = np.zeros(n) if diag else None
    w = 0.9 if mode == "rmsprop" else 0.0
    h0 = 0.37 if mode == "h0" else 0.0
    dev = lambda a: None if a is None else torch.as_tensor(a, device="cuda")
    dx, dg, dS, dY, dG, dH0 = dev(x), dev(g), dev(S), dev(Y), dev(G), dev(H0)
    want = oracle_take_step(0.05, x, g, S, Y, m, used, st_ix, w, H0, h0, G, 1e-4, 1)
    got = hip_take_step(lib, 0.05, dx, dg, dS, dY, m, used, st_ix, w, dH0, h0, dG, 1e-4, 1)
    lib.stochqn_hip_release(C.c_void_p(dS.data_ptr()))
    assert got[:2] == want[:2]
    assert want[0] == 200
    for name, a, b in (("x", dx, x), ("direction", dg, g), ("G", dG, G), ("H0", dH0, H0 if used > 0 else None)):
        if b is not None:
            assert rel_err(a.cpu().numpy(), b) <= TOL, (name, rel_err(a.cpu().numpy(), b))
    if used > 0:
        assert np.allclose(got[2][:used], want[2][:used], rtol=TOL, atol=0)
        assert np.allclose(got[3][:used], want[3][:used], rtol=1e-9, atol=1e-13 * np.abs(want[3][:used]).max())


def test_take_step_guard_rejects_like_the_oracle(form, hip_backend):
    """A non-finite gradient entry: x untouched, memory flushed, search_direction_was_nan -- in both forms."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    n, m = 5000, 4
    rng = np.random.default_rng(3)
    S, Y = make_pairs(rng, n, m)
    g = rng.random(n) - 0.5
    g[n // 3] = np.inf
    x = 1.0 + rng.random(n)
    G, H0 = 0.1 + rng.random(n), np.zeros(n)
    dx, dg, dS, dY, dG, dH0 = (torch.as_tensor(a, device="cuda") for a in (x, g, S, Y, G, H0))
    want = oracle_take_step(0.05, x.copy(), g, S, Y, m, m, 1, 0.9, H0, 0.0, G, 1e-4, 1)
    got = hip_take_step(lib, 0.05, dx, dg, dS, dY, m, m, 1, 0.9, dH0, 0.0, dG, 1e-4, 1)
    lib.stochqn_hip_release(C.c_void_p(dS.data_ptr()))
    assert got[:2] == want[:2] == (203, 0)
    assert np.array_equal(dx.cpu().numpy(), x)


@pytest.mark.parametrize("mode", ["rmsprop", "gamma"])
def test_take_step_host_pointers(mode, form, hip_backend):
    """The same entry fed with plain numpy memory: everything is staged / mirrored behind the ABI and x, the direction,
    G and H0 come back into the caller's arrays."""
    import stochqn_amd
    from stochqn_amd import _abi
    lib = stochqn_amd.cdll()
    take_step_args(lib)
    n, m, used, st_ix = 3001, 5, 5, 2
    rng = np.random.default_rng(12)
    S, Y = make_pairs(rng, n, m)
    g, x = rng.random(n) - 0.5, 1.0 + rng.random(n)
    diag = mode == "rmsprop"
    G = (0.1 + rng.random(n)) if diag else None
    H0 = np.zeros(n) if diag else None
    xw, gw, Gw, H0w = x.copy(), g.copy(), None if G is None else G.copy(), None if H0 is None else H0.copy()
    want = oracle_take_step(0.05, xw, gw, S.copy(), Y.copy(), m, used, st_ix, 0.9 if diag else 0.0, H0w, 0.0, Gw, 1e-4, 1)
    rho, alpha = np.zeros(m), np.zeros(m)
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho.ctypes.data, alpha.ctypes.data, None, None, m, used, st_ix, 1, 0.0, 0.0)
    info = C.c_int(0)
    rc = lib.stochqn_hip_take_step(0.05, n, x.ctypes.data, g.ctypes.data, C.byref(b), 0.9 if diag else 0.0,
                                   None if H0 is None else H0.ctypes.data, 0.0, None if G is None else G.ctypes.data, 1e-4, 1, C.byref(info))
    lib.stochqn_hip_release(C.c_void_p(S.ctypes.data))
    assert rc == 0 and (info.value, b.mem_used) == want[:2]
    for name, a, w in (("x", x, xw), ("direction", g, gw), ("G", G, Gw), ("H0", H0, H0w)):
        if w is not None:
            assert rel_err(a, w) <= TOL, (name, rel_err(a, w))
    assert np.allclose(rho[:used], want[2][:used], rtol=TOL, atol=0)


def test_adaqn_step_matches_the_oracle_at_full_size(hip_backend):
    """adaQN's step at the headline shape, n = 1e8, m = 20, ring full and wrapped, RMSProp diagonal: direction,
    x, G and H0 to 1e-10 against the oracle, for the default three-pass kernels (pass 2 builds H0 and applies the
    side effects) and the sweep form.  (BASELINE config 4's per-step path; the oracle needs 32 GB of host
    memory and a few seconds per call.)"""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    n, m, st_ix = 100_000_000, 20, 3
    d, S, Y, gen = device_pairs(torch, n, m, 20240611)
    g = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    x = 1.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    G = 0.05 + 0.1 * torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    S_h, Y_h = S.cpu().numpy(), Y.cpu().numpy()
    x_w, g_w, G_w, H0_w = x.cpu().numpy(), g.cpu().numpy(), G.cpu().numpy(), np.zeros(n)
    want = oracle_take_step(0.01, x_w, g_w, S_h, Y_h, m, m, st_ix, 0.9, H0_w, 0.0, G_w, 1e-4, 1)
    del S_h, Y_h
    try:
        for which in ("threepass", "sweeps"):
            set_form(lib, which)
            xq, gq, Gq, H0q = x.clone(), g.clone(), G.clone(), torch.zeros_like(g)
            got = hip_take_step(lib, 0.01, xq, gq, S, Y, m, m, st_ix, 0.9, H0q, 0.0, Gq, 1e-4, 1)
            assert got[:2] == want[:2] == (200, m)
            for name, a, b in (("direction", gq, g_w), ("x", xq, x_w), ("G", Gq, G_w), ("H0", H0q, H0_w)):
                e = rel_err(a.cpu().numpy(), b)
                assert e <= TOL, (name, which, e)
            assert np.allclose(got[2], want[2], rtol=TOL, atol=0)
            assert np.allclose(got[3], want[3], rtol=1e-9, atol=1e-13 * np.abs(want[3]).max())
            del xq, gq, Gq, H0q
    finally:
        reset_form(lib)
        lib.stochqn_hip_release(C.c_void_p(S.data_ptr()))


# ---------------------------------------------------------------------------------------------
# empirical Fisher product (reference src/stochqn.c:936-952)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 65, 4096, 100001])
@pytest.mark.parametrize("fu", [1, 7, 8, 9, 32, 100, 500])
def test_fisher_product_matches_oracle(n, fu, hip_backend):
    import stochqn_amd
    from oracle import oracle
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    rng = np.random.default_rng(n + fu)
    F = rng.standard_normal(fu * n)
    s = rng.standard_normal(n)
    t_w, y_w = oracle.fisher_product(F, fu, s)
    dF, ds = torch.as_tensor(F, device="cuda"), torch.as_tensor(s, device="cuda")
    dy = torch.zeros(n, dtype=torch.float64, device="cuda")
    t = np.zeros(fu)
    lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.stochqn_hip_fisher_product(dF.data_ptr(), fu, n, ds.data_ptr(), t.ctypes.data, dy.data_ptr()) == 0
    lib.stochqn_hip_release(C.c_void_p(dF.data_ptr()))
    assert rel_err(t, t_w) <= TOL
    assert rel_err(dy.cpu().numpy(), y_w) <= TOL


def test_fisher_product_matches_the_oracle_at_full_size(hip_backend):
    """n = 1e8 with the 32-row batch of the headline benchmark's Hessian-vector product (25.6 GB of F)."""
    import stochqn_amd
    from oracle import oracle
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    n, fu = 100_000_000, 32
    gen = torch.Generator(device="cuda").manual_seed(5)
    dF = torch.randn(fu * n, dtype=torch.float64, device="cuda", generator=gen)
    ds = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen)
    dy = torch.zeros(n, dtype=torch.float64, device="cuda")
    t = np.zeros(fu)
    lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.stochqn_hip_fisher_product(dF.data_ptr(), fu, n, ds.data_ptr(), t.ctypes.data, dy.data_ptr()) == 0
    lib.stochqn_hip_release(C.c_void_p(dF.data_ptr()))
    t_w, y_w = oracle.fisher_product(dF.cpu().numpy(), fu, ds.cpu().numpy())
    assert rel_err(t, t_w) <= TOL
    assert rel_err(dy.cpu().numpy(), y_w) <= TOL


def _chunked_fisher_reference(torch, F, fu, n, s, chunk=4_000_000):
    """t = F s and y = F't / fu with plain torch fp64 matrix-vector products over column chunks (an independent
    implementation: rocBLAS gemv, other summation order), F = [fu][n] row-major on the device."""
    Fm = F.view(-1, n)[:fu]
    t = torch.zeros(fu, dtype=torch.float64, device=F.device)
    for a in range(0, n, chunk):
        t += Fm[:, a:a + chunk] @ s[a:a + chunk]
    y = torch.empty(n, dtype=torch.float64, device=F.device)
    for a in range(0, n, chunk):
        y[a:a + chunk] = (Fm[:, a:a + chunk].t() @ t) / fu
    return t, y


@pytest.fixture(scope="class")
def c4_run(hip_backend):
    """BASELINE config 4 as stated: adaQN, n = 1e8, m = 20, fisher_size = 128 (102.4 GB ring), RMSProp H0, L = 20,
    min_curvature = 1e-4.  141 iterations fill the Fisher ring; stops right after the call that built the pair of iteration 140."""
    torch = torch_cuda()
    n, m, f, L = 100_000_000, 20, 128, 20
    gen = torch.Generator(device="cuda").manual_seed(4)
    d = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    dn = [d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 1)) for _ in range(3)]
    x = 1 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    opt = OPTIMIZERS["adaQN"](backend=hip_backend, space="device", mem_size=m, fisher_size=f, bfgs_upd_freq=L, max_incr=None,
                              min_curvature=1e-4, rmsprop_weight=0.9, scal_reg=1e-4)
    t, infos = 0, set()
    while (opt.niter if opt.initialized else 0) < 141:
        r = opt.run_optimizer(x, 0.01)
        infos.add(r["info"]["iteration_info"])
        assert r["task"] == "calc_grad"
        torch.mul(dn[t % 3], r["requested_on"], out=opt.gradient)
        t += 1
        if opt.niter == 140 and opt.section == 1 and opt.Fisher_mem.mem_used == f:
            break                                               # right after the call that built the pair of iteration 140
    del d, dn, x
    row = (opt.BFGS_mem.mem_st_ix - 1) % m
    yield {"opt": opt, "n": n, "m": m, "f": f, "infos": infos,
           "s": opt.BFGS_mem.s_mem[row * n:(row + 1) * n], "y": opt.BFGS_mem.y_mem[row * n:(row + 1) * n]}
    opt.release()


class TestAdaqnAtTheC4Shape:
    def test_adaqn_at_the_c4_shape_builds_its_pairs_from_all_128_fisher_rows(self, c4_run):
        """The pair built at iteration 140 must be y = F'(F s)/128 over ALL 128 rows -- checked on the device against chunked
        torch fp64 products (an independent implementation) -- and satisfy s'y = |F s|^2 / 128."""
        torch = torch_cuda()
        opt, n, m, f, s, y = (c4_run[k] for k in ("opt", "n", "m", "f", "s", "y"))
        assert c4_run["infos"] == {"no_problems_encountered"}
        assert opt.Fisher_mem.mem_used == f and opt.BFGS_mem.mem_used == 6          # pairs at 40, 60, ..., 140
        t_ref, y_ref = _chunked_fisher_reference(torch, opt.Fisher_mem.F, f, n, s)
        err = float(torch.linalg.norm(y - y_ref) / torch.linalg.norm(y_ref))
        assert err <= TOL, err
        assert rel_err(opt.Fisher_mem.buffer_y, t_ref.cpu().numpy()) <= TOL
        sy, tt = float(torch.dot(s, y)), float(torch.dot(t_ref, t_ref)) / f
        assert abs(sy - tt) <= 1e-9 * abs(tt), (sy, tt)
        assert float(torch.linalg.norm(s)) > 0

    def test_the_128_row_pair_of_the_c4_shape_matches_the_oracle(self, c4_run):
        """The same pair against the ORACLE (reference src/stochqn.c:936-952 restated): the whole 102.4 GB Fisher ring is
        brought to the host (the GPU boxes have ~260 GB of it; skipped, visibly, where the host is smaller), the oracle forms
        t = F s and y = F't/128 from the library's own s, and t (buffer_y) and y must agree to 1e-10 -- C4's pair at C4's size."""
        from oracle import oracle
        torch = torch_cuda()
        opt, n, f, s, y = (c4_run[k] for k in ("opt", "n", "f", "s", "y"))
        need = (f + 4) * n * 8
        avail = 0
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) * 1024
        try:
            lim = open("/sys/fs/cgroup/memory.max").read().strip()
            if lim != "max":
                avail = min(avail, int(lim) - int(open("/sys/fs/cgroup/memory.current").read()))
        except (OSError, ValueError):
            pass
        if avail < 1.35 * need:
            pytest.skip("the host has %.0f GB available, the 128-row Fisher ring needs %.0f GB there" % (avail / 1e9, 1.35 * need / 1e9))
        olib = oracle.cdll()
        oracle.set_threads(oracle.usable_cpus())
        F_h = np.empty(f * n)
        if hasattr(olib, "oracle_first_touch"):
            olib.oracle_first_touch(F_h.ctypes.data, f * n)
        for k in range(f):                                                # device -> the numpy buffer itself, row by row
            torch.from_numpy(F_h[k * n:(k + 1) * n]).copy_(opt.Fisher_mem.F[k * n:(k + 1) * n])
        s_h = s.cpu().numpy()
        t_w, y_w = np.zeros(f), np.empty(n)
        olib.oracle_fisher_product(F_h.ctypes.data, f, n, s_h.ctypes.data, t_w.ctypes.data, y_w.ctypes.data)
        del F_h
        e_y, e_t = rel_err(y.cpu().numpy(), y_w), rel_err(np.asarray(opt.Fisher_mem.buffer_y), t_w)
        print("C4 pair at fisher_size = 128, n = 1e8 against the oracle: y %.2e, t %.2e" % (e_y, e_t))
        assert e_y <= TOL and e_t <= TOL, (e_y, e_t)


def test_adaqn_at_the_c4_shape_first_cycle_matches_the_oracle(hip_backend, oracle_backend):
    """BASELINE config 4 AS STATED -- adaQN, n = 1e8, m = 20, fisher_size = 128, L = 20, RMSProp 0.9, scal_reg 1e-4,
    min_curvature 1e-4 -- from its first call through one whole correction-pair cycle against the oracle: forty rescaled-gradient
    steps that fill forty rows of the Fisher ring, the average archived at iteration 20, the pair of iteration 40 built from those
    forty rows (y = F'(F s)/40, reference src/stochqn.c:936-952, accepted against min_curvature), and the step after it, which is
    the first to use a pair and the diagonal H0 together.  Identical task / info / counter sequences; x, G, H0, the averages,
    the pair and the Fisher rows at 1e-10.  (The 128-row pair is TestAdaqnAtTheC4Shape's; a whole trajectory with a pair per
    iteration is the lock-step test's.)  The oracle's Fisher ring is 102 GB of untouched zero pages but for its 41 rows."""
    torch = torch_cuda()
    n, m, f, L, iters, step = 100_000_000, 20, 128, 20, 41, 0.01
    need = (41 + 2 + 12) * n * 8                    # touched: 41 Fisher rows, one pair, the n-vectors and three gradient factors
    avail = _host_memory_available()
    if avail < 1.3 * need:
        pytest.skip("the host has %.0f GB available, the oracle's side of the C4 cycle needs %.0f GB" % (avail / 1e9, 1.3 * need / 1e9))
    kw = dict(mem_size=m, fisher_size=f, bfgs_upd_freq=L, max_incr=None, min_curvature=1e-4, rmsprop_weight=0.9, scal_reg=1e-4)
    gen = torch.Generator(device="cuda").manual_seed(4)
    d = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    dn_d = [d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 1)) for _ in range(3)]
    x_d = 1 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    del d
    dn_h, x_h = [to_np(a) for a in dn_d], to_np(x_d)
    ref = OPTIMIZERS["adaQN"](backend=oracle_backend, space="host", **kw)
    dev = OPTIMIZERS["adaQN"](backend=hip_backend, space="device", **kw)
    try:
        t = 0
        while (ref.niter if ref.initialized else 0) < iters:
            r_r, r_d = ref.run_optimizer(x_h, step), dev.run_optimizer(x_d, step)
            state = lambda o: (o.niter, o.section, o.BFGS_mem.mem_used, o.BFGS_mem.mem_st_ix, o.Fisher_mem.mem_used, o.Fisher_mem.mem_st_ix)
            assert (r_d["task"], r_d["info"], state(dev)) == (r_r["task"], r_r["info"], state(ref)), (t, r_d["task"], r_r["task"], state(dev), state(ref))
            assert r_r["task"] == "calc_grad"
            np.multiply(dn_h[t % 3], r_r["requested_on"], out=ref.gradient)
            torch.mul(dn_d[t % 3], r_d["requested_on"], out=dev.gradient)
            t += 1
        assert ref.BFGS_mem.mem_used == 1 and ref.Fisher_mem.mem_used == 41 and ref.niter == 41

        def close(what, got, want):
            wd = torch.from_numpy(np.ascontiguousarray(want)).to("cuda")
            e = float(torch.linalg.vector_norm(got - wd) / torch.linalg.vector_norm(wd))
            print("C4 first cycle, %-14s %.2e from the oracle's" % (what + ":", e))
            assert e <= TOL, (what, e)

        close("x", x_d, x_h)
        close("grad_sum_sq", dev.grad_sum_sq, ref.grad_sum_sq)
        close("H0", dev.H0, ref.H0)
        close("x_avg_prev", dev.x_avg_prev, ref.x_avg_prev)
        close("s of the pair", dev.BFGS_mem.s_mem[:n], ref.BFGS_mem.s_mem[:n])
        close("y of the pair", dev.BFGS_mem.y_mem[:n], ref.BFGS_mem.y_mem[:n])
        for row in (0, 20, 40):
            close("Fisher row %d" % row, dev.Fisher_mem.F[row * n:(row + 1) * n], ref.Fisher_mem.F[row * n:(row + 1) * n])
        assert rel_err(np.asarray(dev.Fisher_mem.buffer_y)[:40], np.asarray(ref.Fisher_mem.buffer_y)[:40]) <= TOL
    finally:
        dev.release()
        ref.release()


# ---------------------------------------------------------------------------------------------
# size-independent properties at BASELINE sizes (the oracle is too slow / too big there)
# ---------------------------------------------------------------------------------------------
def device_pairs(torch, n, m, seed):
    """d_i, S, Y = d*S generated on the device (SURVEY.md 8d synthetic inputs, torch RNG)."""
    gen = torch.Generator(device="cuda").manual_seed(seed)
    d = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    S = torch.empty(m * n, dtype=torch.float64, device="cuda")
    Y = torch.empty(m * n, dtype=torch.float64, device="cuda")
    for k in range(m):
        s = 1e-3 * (torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5)
        S[k * n:(k + 1) * n] = s
        Y[k * n:(k + 1) * n] = d * s
    return d, S, Y, gen


@pytest.mark.parametrize("n,m", [(10_000_000, 10), (100_000_000, 20), (125_000_000, 20)])      # C2, C3 / C4, C5's per-GPU shard
def test_two_loop_properties_at_baseline_size(n, m, form, hip_backend):
    """(i) secant equation: the L-BFGS inverse maps the newest y onto the newest s exactly;
    (ii) linearity in the gradient; (iii) bit-reproducibility of a repeated call."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    d, S, Y, gen = device_pairs(torch, n, m, 20240611)
    st = 3 % m
    newest = (st + m - 1) % m
    q = Y[newest * n:(newest + 1) * n].clone()
    hip_two_loop(lib, q, None, 0.0, Y, S, n, m, m, st)
    s_new = S[newest * n:(newest + 1) * n]
    assert float(torch.linalg.norm(q - s_new) / torch.linalg.norm(s_new)) <= TOL

    g1 = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    g2 = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    a, b = 0.75, -1.25
    mix = a * g1 + b * g2
    r1, r2 = g1.clone(), g2.clone()
    hip_two_loop(lib, r1, None, 0.0, Y, S, n, m, m, st)
    hip_two_loop(lib, r2, None, 0.0, Y, S, n, m, m, st)
    hip_two_loop(lib, mix, None, 0.0, Y, S, n, m, m, st)
    lin = a * r1 + b * r2
    assert float(torch.linalg.norm(mix - lin) / torch.linalg.norm(lin)) <= TOL

    again = g1.clone()
    hip_two_loop(lib, again, None, 0.0, Y, S, n, m, m, st)
    assert torch.equal(again, r1)
    lib.stochqn_hip_release(C.c_void_p(S.data_ptr()))


def test_two_loop_matches_the_oracle_at_full_size(hip_backend):
    """The headline shape itself against the oracle: n = 1e8, m = 20, ring full and wrapped, fp64.  The oracle
    needs the 32 GB of S and Y in host memory and ~4 s per two-loop on the box's 16 CPUs, so this is done once,
    for both forms of the recursion, with the scalar H0 and with a caller-supplied diagonal.  adaQN's step at this size:
    test_adaqn_step_matches_the_oracle_at_full_size."""
    import stochqn_amd
    from oracle import oracle
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    n, m, st = 100_000_000, 20, 3
    d, S, Y, gen = device_pairs(torch, n, m, 20240611)
    g = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    H0 = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    S_h, Y_h = S.cpu().numpy(), Y.cpu().numpy()
    try:
        for H0_d in (None, H0):
            want = g.cpu().numpy().copy()
            rho_w, alpha_w = oracle.two_loop(want, None if H0_d is None else H0_d.cpu().numpy(), 0.0, Y_h, S_h, m, m, st)
            for which in ("threepass", "sweeps"):
                set_form(lib, which)
                q = g.clone()
                rho, alpha = hip_two_loop(lib, q, H0_d, 0.0, Y, S, n, m, m, st)
                got = q.cpu().numpy()
                assert rel_err(got, want) <= TOL, (H0_d is not None, which, rel_err(got, want))
                assert np.allclose(rho, rho_w, rtol=TOL, atol=0)
                assert np.allclose(alpha, alpha_w, rtol=1e-9, atol=1e-13 * np.abs(alpha_w).max())
                del q, got
    finally:
        reset_form(lib)
        lib.stochqn_hip_release(C.c_void_p(S.data_ptr()))


@pytest.mark.parametrize("optname,n,kw,iters,step,tol", [
    ("SQN", 100_000_000, dict(mem_size=20, bfgs_upd_freq=1, min_curvature=None), 23, 0.05, TOL),
    ("oLBFGS", 10_000_000, dict(mem_size=10, min_curvature=None), 40, 0.05, TOL),           # the C2 shape
])                                                                                          # (adaQN: the lock-step test below)
def test_steps_match_the_oracle_at_full_size(optname, n, kw, iters, step, tol, hip_backend, oracle_backend):
    """Whole optimiser steps at the BASELINE shapes against the oracle: iterations from the same start with a
    new pair every iteration (ring filling up and wrapping), identical gradients and Hessian-vector products fed
    to both (element-wise products, bit-identical in numpy and torch).  Final x to the tolerance, every discrete
    output identical.  The oracle side costs ~15 s on the box's 16 CPUs and 32 GB of host memory at n = 1e8."""
    torch = torch_cuda()
    gen = torch.Generator(device="cuda").manual_seed(99)
    d_d = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    dn_d = [d_d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 1)) for _ in range(2)]
    x0_d = 1 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    d_h, dn_h, x0_h = d_d.cpu().numpy(), [a.cpu().numpy() for a in dn_d], x0_d.cpu().numpy()

    def run(backend, space, d, dn, x, mul):
        opt = OPTIMIZERS[optname](backend=backend, space=space, **kw)
        log, t = [], 0
        while (opt.niter if opt.initialized else 0) < iters:
            r = opt.run_optimizer(x, step)
            log.append((r["task"], r["info"]["iteration_info"], opt.niter, opt.BFGS_mem.mem_used, opt.BFGS_mem.mem_st_ix))
            if r["task"] == "calc_grad":
                mul(dn[t % 2], r["requested_on"], opt.gradient)
                t += 1
            elif r["task"] == "calc_grad_same_batch":                  # oLBFGS: same noise as the last gradient
                mul(dn[(t - 1) % 2], r["requested_on"], opt.gradient)
            elif r["task"] == "calc_hess_vec":
                mul(d, r["requested_on"][1], opt.hess_vec)
        opt.release()
        return x, log

    x_ref, log_ref = run(oracle_backend, "host", d_h, dn_h, x0_h.copy(), lambda a, b, out: np.multiply(a, b, out=out))
    x_dev, log_dev = run(hip_backend, "device", d_d, dn_d, x0_d.clone(), lambda a, b, out: torch.mul(a, b, out=out))
    assert log_dev == log_ref
    assert log_ref[-1][3] == kw["mem_size"]
    assert rel_err(x_dev.cpu().numpy(), x_ref) <= tol
    assert rel_err(x_ref, x0_h) > 1e-4


def test_adaqn_trajectory_at_full_size_in_lockstep_with_the_oracle(hip_backend, oracle_backend):
    """A whole adaQN trajectory at the C4 size (n = 1e8, m = 20, Fisher pairs, RMSProp diagonal) held to the oracle at
    1e-10.  adaQN's Fisher pairs can amplify last-bit differences (FREE_RUN_TOL above: up to 1e-7 between two summation orders
    of the ORACLE on one of the small configurations); here the device state is put back on the oracle's every K = 5 iterations -- x, G, H0, the
    averages and the rows of S, Y and F written since the last sync: n-vectors, not the rings -- so the amplification cannot
    accumulate, and at every sync point x, G, H0 and those rows are compared at the north-star tolerance.  A second device
    optimiser runs free beside them and ends within 1e-10 of the oracle too (measured 3.3e-16: over these 22 iterations nothing is
    amplified yet; until round 6 the bar was 1e-7).  22 iterations: the ring of 20
    fills and wraps, the Fisher ring of 16 wraps.  Reference: src/stochqn.c:1170-1239 (the step), :936-952 (the pair)."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    n, iters, step, K = 100_000_000, 22, 0.002, int(os.environ.get("SQN_TEST_K", "5"))
    kw = dict(mem_size=20, fisher_size=16, bfgs_upd_freq=1, max_incr=None, min_curvature=None, rmsprop_weight=0.9)
    gen = torch.Generator(device="cuda").manual_seed(99)
    d_d = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    dn_d = [d_d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 1)) for _ in range(2)]
    x0_d = 1 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    dn_h, x0_h = [a.cpu().numpy() for a in dn_d], x0_d.cpu().numpy()
    del d_d
    ref = OPTIMIZERS["adaQN"](backend=oracle_backend, space="host", **kw)
    lock = OPTIMIZERS["adaQN"](backend=hip_backend, space="device", **kw)
    free = OPTIMIZERS["adaQN"](backend=hip_backend, space="device", **kw)
    x_ref, x_lock, x_free = x0_h.copy(), x0_d.clone(), x0_d.clone()
    t, syncs, worst = 0, 0, 0.0

    bound = {}                                      # ring row -> what the cancellation in s = x_avg - x_avg_prev allows

    def close(what, got, want):
        """x, G, H0, the averages and the Fisher rows at 1e-10 (measured: 1e-17 -- they are bit-identical but for single ulps).
        s = x_avg - x_avg_prev is a difference of two vectors that agree to 7-8 digits at this step size (|s| / |x| ~ 3e-8):
        one ulp of x is eps |x| / |s| ~ 4e-9 of s in ANY fp64 evaluation, the reference's included, and y = F'(F s)/fu is
        linear in s.  So the rows of S and Y are held to what that allows, 16 eps |x| / |s| (measured: up to 1.1e-9, at the
        last sync point), and -- like everything else here -- to 1e-10 of the vectors they were computed FROM (|x|)."""
        nonlocal worst
        w = torch.from_numpy(np.ascontiguousarray(want)).to("cuda")          # compared where the 0.8 GB vectors are: on the device
        assert bool(torch.isfinite(w).all()) and bool(torch.isfinite(got).all()), what
        nw = float(torch.linalg.vector_norm(w))
        e = float(torch.linalg.vector_norm(got - w)) / nw if nw > 0 else float(torch.linalg.vector_norm(got))
        tol = TOL
        if what.startswith(("s_mem", "y_mem")):
            row = int(what.split()[-1])
            if what.startswith("s_mem"):
                nx = float(np.linalg.norm(x_ref))
                bound[row] = max(TOL, 16 * np.finfo(np.float64).eps * nx / nw)
                assert e * nw <= TOL * nx, (what, e)                         # absolute error against |x|: 1e-10
            tol = bound[row]
        else:
            worst = max(worst, e)
        if os.environ.get("SQN_TEST_REPORT_ONLY"):
            print("iteration %d: %-16s %.3e (held to %.1e)" % (ref.niter, what, e, tol))
            return w
        assert e <= tol, "iteration %d: %s is %.3e from the oracle's (held to %.1e)" % (ref.niter, what, e, tol)
        return w

    def rows(mem, size, st, used, name):
        a_r, a_l = getattr(getattr(ref, mem), name), getattr(getattr(lock, mem), name)
        for j in range(min(K, used)):
            r = (st - 1 - j) % size
            yield "%s row %d" % (name, r), a_l[r * n:(r + 1) * n], a_r[r * n:(r + 1) * n]

    try:
        while (ref.niter if ref.initialized else 0) < iters:
            rs = [o.run_optimizer(x, step) for o, x in ((ref, x_ref), (lock, x_lock), (free, x_free))]
            assert rs[0]["task"] == rs[1]["task"] == rs[2]["task"] == "calc_grad" and rs[0]["info"] == rs[1]["info"] == rs[2]["info"]
            for o in (lock, free):
                assert (o.niter, o.section, o.BFGS_mem.mem_used, o.BFGS_mem.mem_st_ix, o.Fisher_mem.mem_used, o.Fisher_mem.mem_st_ix) == \
                       (ref.niter, ref.section, ref.BFGS_mem.mem_used, ref.BFGS_mem.mem_st_ix, ref.Fisher_mem.mem_used, ref.Fisher_mem.mem_st_ix)
            if ref.niter > 0 and ref.niter % K == 0:                       # a sync point: compare at 1e-10, then put the device state on the oracle's
                pieces = [("x", x_lock, x_ref), ("G", lock.grad_sum_sq, ref.grad_sum_sq), ("H0", lock.H0, ref.H0),
                          ("x_sum", lock.x_sum, ref.x_sum), ("x_avg_prev", lock.x_avg_prev, ref.x_avg_prev)]
                pieces += list(rows("BFGS_mem", 20, ref.BFGS_mem.mem_st_ix, ref.BFGS_mem.mem_used, "s_mem"))
                pieces += list(rows("BFGS_mem", 20, ref.BFGS_mem.mem_st_ix, ref.BFGS_mem.mem_used, "y_mem"))
                pieces += list(rows("Fisher_mem", 16, ref.Fisher_mem.mem_st_ix, ref.Fisher_mem.mem_used, "F"))
                for what, got, want in pieces:
                    if what not in ("x_sum",):                               # x_sum is zero between the iterations at L = 1
                        got.copy_(close(what, got, want))                    # ... and put on the oracle's, from the copy that is up already
                    else:
                        lock._sp.assign(got, want)
                lib.stochqn_hip_invalidate(C.c_void_p(lock._sp.ptr(lock.BFGS_mem.s_mem)))
                syncs += 1
            np.multiply(dn_h[t % 2], rs[0]["requested_on"], out=ref.gradient)
            torch.mul(dn_d[t % 2], rs[1]["requested_on"], out=lock.gradient)
            torch.mul(dn_d[t % 2], rs[2]["requested_on"], out=free.gradient)
            t += 1
        assert syncs == iters // K and ref.BFGS_mem.mem_used == 20 and ref.Fisher_mem.mem_used == 16
        close("the final x", x_lock, x_ref)
        x_ref_d = to_dev(x_ref)
        e_free = float(torch.linalg.vector_norm(x_free - x_ref_d) / torch.linalg.vector_norm(x_ref_d))
        assert e_free <= TOL, e_free                                         # free-running over these 22 iterations: measured 3.3e-16 (round 6), held to 1e-10 (was 1e-7)
        assert rel_err(x_ref, x0_h) > 1e-4                                   # and the run went somewhere
        print("adaQN at n = 1e8 in lock-step every %d iterations: worst relative error of x / G / H0 / F at a sync point %.2e; free-running %.2e" % (K, worst, e_free))
    finally:
        for o in (lock, free, ref):
            o.release()


@pytest.mark.parametrize("optname,kw,iters,step,tol", [
    ("SQN", dict(mem_size=20, bfgs_upd_freq=1, min_curvature=None), 30, 0.05, TOL),
    ("oLBFGS", dict(mem_size=20, min_curvature=None), 30, 0.05, TOL),
    # adaQN's Fisher pairs make the free-running trajectory amplify last-bit differences: after these 26 iterations at n = 1e8 two
    # summation orders of the ORACLE's dot products end 1.2e-9 apart (profiles/r06_oracle_vs_oracle_sensitivity.json:
    # full_size_n_1e8), the two forms on the device 2.8e-9: held to 2e-8 (was 1e-7)
    ("adaQN", dict(mem_size=20, fisher_size=16, bfgs_upd_freq=1, max_incr=None, min_curvature=None, rmsprop_weight=0.9),
     26, 0.002, 2e-8),
])
def test_full_size_steps_agree_between_the_two_forms(optname, kw, iters, step, tol, hip_backend):
    """n = 1e8, m = 20 (BASELINE size), whole optimiser steps: the three-pass form and the chain of
    dependent sweeps are two independent implementations of the same recursion, each held to the oracle at
    test sizes; here they are held to each other where the oracle is too slow -- 30 iterations from the
    same start (ring filling up, then wrapping), x equal to 1e-10, every discrete output identical."""
    import gc
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    n = 100_000_000
    gen = torch.Generator(device="cuda").manual_seed(20240611)
    d = 0.5 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    dn = d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 1))
    x0 = 1 + torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)

    def run(which):
        set_form(lib, which)
        opt = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
        x = x0.clone()
        log = []
        while (opt.niter if opt.initialized else 0) < iters:
            r = opt.run_optimizer(x, step)
            log.append((r["task"], r["info"]["iteration_info"], opt.niter, opt.BFGS_mem.mem_used, opt.BFGS_mem.mem_st_ix))
            if r["task"] in ("calc_grad", "calc_grad_same_batch"):
                torch.mul(dn if r["task"] == "calc_grad" else d, r["requested_on"], out=opt.gradient)
            elif r["task"] == "calc_hess_vec":
                torch.mul(d, r["requested_on"][1], out=opt.hess_vec)
        opt.release()
        del opt
        gc.collect()
        torch.cuda.empty_cache()
        return x, log

    try:
        xa, la = run("threepass")
        xb, lb = run("sweeps")
    finally:
        reset_form(lib)
    assert la == lb
    assert la[-1][3] == 20                                     # the ring did fill up
    err = float(torch.linalg.norm(xa - xb) / torch.linalg.norm(xb))
    print("%s at n = 1e8, %d iterations: the two forms of the recursion end %.2e apart (held to %.1e)" % (optname, iters, err, tol))
    assert err <= tol, err
    assert float(torch.linalg.norm(xa - x0) / torch.linalg.norm(x0)) > 1e-4     # and the run went somewhere


# ---------------------------------------------------------------------------------------------
# state that lives behind the ABI: export / resume, the bak->slot quirk with a full ring, options
# ---------------------------------------------------------------------------------------------
def _lib():
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    lib.stochqn_hip_export.argtypes = [C.c_void_p]
    lib.stochqn_hip_release.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("optname,kw", [
    ("SQN", dict(mem_size=3, bfgs_upd_freq=4)),
    ("oLBFGS", dict(mem_size=4)),
    ("adaQN", dict(mem_size=3, fisher_size=6, bfgs_upd_freq=4, rmsprop_weight=0.9)),
])
def test_export_and_resume_host_state(optname, kw, form, hip_backend, oracle_backend):
    """Host caller (profile B): export refreshes every host array from its device mirror, so that
    the object can be pickled; dropping the context (= new process) and calling on re-imports it."""
    lib = _lib()
    n = 777
    P = NoisyQuadratic(n, seed=5)
    ref = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
    x_ref, x = P.x0(), P.x0()
    want = run_trace(ref, P, x_ref, 0.05, 41)
    got = run_trace(opt, P, x, 0.05, 41)
    compare_traces(got, want, TOL)
    key = C.c_void_p(opt.BFGS_mem.s_mem.ctypes.data)
    assert lib.stochqn_hip_export(key) == 0
    used = opt.BFGS_mem.mem_used
    assert used > 0
    for name in ("s_mem", "y_mem"):
        a, b = getattr(opt.BFGS_mem, name), getattr(ref.BFGS_mem, name)
        assert rel_err(a, b) <= TOL, name
    for name in ("x_sum", "x_avg_prev", "grad_sum_sq", "grad_prev"):
        if hasattr(ref, name) and getattr(ref, name).shape[0] == n:
            assert rel_err(getattr(opt, name), getattr(ref, name)) <= TOL, name
    if hasattr(ref, "Fisher_mem"):
        assert rel_err(opt.Fisher_mem.F, ref.Fisher_mem.F) <= TOL
    lib.stochqn_hip_release(key)                       # "new process": the device context is gone
    # both continue from their host state; call indices keep counting so the noise stays in step
    P2 = NoisyQuadratic(n, seed=6)
    want2 = run_trace(ref, P2, x_ref, 0.05, 40)
    got2 = run_trace(opt, P2, x, 0.05, 40)
    compare_traces(got2, want2, TOL)


@pytest.mark.parametrize("optname,kw", [
    ("SQN", dict(mem_size=3, bfgs_upd_freq=4)),
    ("oLBFGS", dict(mem_size=4)),
    ("adaQN", dict(mem_size=3, fisher_size=6, bfgs_upd_freq=4, rmsprop_weight=0.9)),
])
def test_pickle_round_trip_of_a_running_host_object(optname, kw, form, hip_backend, oracle_backend):
    """The reference's Python objects pickle correctly because they ARE their state.  Here the arrays of a
    host-space object live in HBM between calls: __getstate__ exports them first (stochqn_hip_export), and the
    copy -- new arrays at new addresses, a new device context -- continues exactly like the oracle."""
    import copy
    import pickle
    n = 600
    P = NoisyQuadratic(n, seed=5)
    ref = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
    x_ref, x = P.x0(), P.x0()
    compare_traces(run_trace(opt, P, x, 0.05, 37), run_trace(ref, P, x_ref, 0.05, 37), TOL)
    clone = pickle.loads(pickle.dumps(opt))
    assert rel_err(clone.BFGS_mem.s_mem, ref.BFGS_mem.s_mem) <= TOL and rel_err(clone.BFGS_mem.y_mem, ref.BFGS_mem.y_mem) <= TOL
    assert (clone.niter, clone.section, clone.BFGS_mem.mem_used) == (ref.niter, ref.section, ref.BFGS_mem.mem_used)
    twin = copy.deepcopy(opt)
    P2 = NoisyQuadratic(n, seed=6)
    want = run_trace(ref, P2, x_ref, 0.05, 30)
    for obj in (clone, twin, opt):                                   # the original keeps working as well
        # run_trace fed the last request before it stopped: the gradient / Hessian-vector product / f it stored travels along
        compare_traces(run_trace(obj, P2, x.copy(), 0.05, 30), want, TOL)
    dev = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
    dev.run_optimizer(torch_cuda().as_tensor(P.x0(), device="cuda"), 0.05)
    with pytest.raises(TypeError):
        pickle.dumps(dev)


def test_device_caller_on_a_side_stream_is_ordered(hip_backend, oracle_backend):
    """The library's stream waits for the NULL stream only; free.py drains the caller's current stream when it
    is another one, so a gradient produced on a side stream is complete before the step reads it."""
    torch = torch_cuda()
    n = 2_000_000
    P = NoisyQuadratic(n, seed=3)
    kw = dict(mem_size=3, bfgs_upd_freq=3)
    want = run_trace(OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw), P, P.x0(), 0.05, 25)
    opt = OPTIMIZERS["SQN"](backend=hip_backend, space="device", **kw)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        x = torch.as_tensor(P.x0(), device="cuda")
        got = run_trace(opt, P, x, 0.05, 25)                      # update_gradient's H2D copies run on `side`
    compare_traces(got, want, TOL)
    opt.release()


def _edit_pairs_in_place(opt, n, to_t):
    """Row 0 of the ring gets a tenth of row 1 added, in S and in Y (stays a valid pair of the quadratic)."""
    for name in ("s_mem", "y_mem"):
        a = getattr(opt.BFGS_mem, name)
        a[0:n] += 0.1 * a[n:2 * n]


@pytest.mark.parametrize("invalidate", [True, False])
def test_in_place_edit_of_the_ring_by_a_device_caller(invalidate, form, hip_backend, oracle_backend, capfd):
    """The cache contract for device callers: the library learns about changes to S / Y only
    through its own writes or stochqn_hip_invalidate.  With the call, an in-place edit continues exactly like
    the oracle on the same edited arrays; without it, option verify_cache turns the silent wrong direction into
    a failed call."""
    lib = _lib()
    n = 1500
    P = NoisyQuadratic(n, seed=8)
    kw = dict(mem_size=3, bfgs_upd_freq=2, min_curvature=None)
    ref = OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS["SQN"](backend=hip_backend, space="device", **kw)
    x_ref, x = P.x0(), torch_cuda().as_tensor(P.x0(), device="cuda")
    compare_traces(run_trace(opt, P, x, 0.05, 31), run_trace(ref, P, x_ref, 0.05, 31), TOL)
    assert ref.BFGS_mem.mem_used == 3
    _edit_pairs_in_place(ref, n, None)
    _edit_pairs_in_place(opt, n, None)
    key = C.c_void_p(opt.BFGS_mem.s_mem.data_ptr())
    P2 = NoisyQuadratic(n, seed=9)
    if invalidate:
        lib.stochqn_hip_invalidate(key)
        compare_traces(run_trace(opt, P2, x, 0.05, 24), run_trace(ref, P2, x_ref, 0.05, 24), TOL)
    else:
        assert lib.stochqn_hip_set_option(b"verify_cache", 1.0) == 0
        try:
            with pytest.raises(ValueError):
                run_trace(opt, P2, x, 0.05, 12)             # round robin over the 3 pairs in use: row 0 within 3 steps
        finally:
            lib.stochqn_hip_set_option(b"verify_cache", 0.0)
        assert "verify_cache: the cached inner products of ring row 0" in capfd.readouterr().err
    opt.release()


def test_verify_cache_is_silent_on_untouched_state(form, hip_backend, oracle_backend):
    """The debugging option must not change results nor raise false alarms (all three optimisers, ring wrapping)."""
    lib = _lib()
    assert lib.stochqn_hip_set_option(b"verify_cache", 1.0) == 0
    try:
        for name in ("sqn_hessvec", "olbfgs_default", "adaqn_fisher_rms", "sqn_ring20", "sqn_reject"):
            cfg = [c for c in CONFIGS if c[0] == name][0]
            got, want = both_traces(cfg, 1000, "device", hip_backend, oracle_backend)
            compare_traces(got, want, FREE_RUN_TOL.get(name, TOL))
    finally:
        lib.stochqn_hip_set_option(b"verify_cache", 0.0)


def test_rejected_pair_with_full_ring_reproduces_bak_quirk(form, hip_backend, oracle_backend):
    """once the ring is full, a rejected pair leaves the bak buffers' contents in
    the slot of the oldest (still counted) pair; with zero bak buffers the next direction is NaN and
    the ring is flushed.  min_curvature is raised mid-run (a field callers may change at any time)."""
    import stochqn_amd
    n = 500
    P = NoisyQuadratic(n, seed=9)
    kw = dict(mem_size=3, min_curvature=1e-6)
    ref = OPTIMIZERS["oLBFGS"](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS["oLBFGS"](backend=hip_backend, space="device", **kw)
    x_ref = P.x0()
    x_dev = torch_cuda().as_tensor(P.x0(), device="cuda")
    lib = stochqn_amd.cdll()
    inval = lambda o: lib.stochqn_hip_invalidate(C.c_void_p(o._sp.ptr(o.BFGS_mem.s_mem)))
    run_lockstep(ref, opt, P, x_ref, x_dev, 0.1, 13, TOL, on_sync=inval)       # ring full (3 pairs)
    assert ref.BFGS_mem.mem_used == 3
    ref.BFGS_mem.min_curvature = opt.BFGS_mem.min_curvature = 1e6              # reject everything from now on
    infos = []

    class Spy:
        def __init__(self, o): self.o = o
        def __getattr__(self, k): return getattr(self.o, k)
        def run_optimizer(self, x, s):
            r = self.o.run_optimizer(x, s)
            infos.append(r["info"]["iteration_info"])
            return r
    run_lockstep(Spy(ref), opt, P, x_ref, x_dev, 0.1, 12, TOL, on_sync=inval)
    assert "curvature_too_small" in infos and "search_direction_was_nan" in infos


def test_strict_grad_option_host_caller(hip_backend, oracle_backend):
    """strict_grad=0: a host caller's `grad` array is left alone (no PCIe copy of the direction);
    x and the requests are unaffected."""
    lib = _lib()
    n = 300
    P = NoisyQuadratic(n, seed=2)
    kw = dict(mem_size=3, bfgs_upd_freq=3)
    want = run_trace(OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw), P, P.x0(), 0.1, 30)
    try:
        assert lib.stochqn_hip_set_option(b"strict_grad", 0.0) == 0
        opt = OPTIMIZERS["SQN"](backend=hip_backend, space="host", **kw)
        x = P.x0()
        last = None
        got = []
        for call in range(30):
            r = opt.run_optimizer(x, 0.1)
            if last is not None:
                assert np.array_equal(opt.gradient, last)          # untouched by the library
            rec = {"task": r["task"], "info": r["info"]["iteration_info"], "x": x.copy(), "niter": opt.niter}
            got.append(rec)
            if r["task"] == "calc_hess_vec":
                rx, rv = r["requested_on"]
                opt.update_hess_vec(P.hess_vec(rx.copy(), rv.copy()))
            else:
                opt.update_gradient(P.grad(np.asarray(r["requested_on"]).copy(), call))
            last = opt.gradient.copy()
        for g, w in zip(got, want):
            assert g["task"] == w["task"] and g["info"] == w["info"] and g["niter"] == w["niter"]
            assert rel_err(g["x"], w["x"]) <= TOL
    finally:
        lib.stochqn_hip_set_option(b"strict_grad", 1.0)


def test_single_rank_rccl_path_matches(form, hip_backend, oracle_backend):
    """With a communicator attached every reduction goes k_fin -> ncclAllReduce -> consumer; with
    one rank the numbers must not change."""
    import stochqn_amd
    lib = stochqn_amd.cdll()
    buf = (C.c_ubyte * 128)()
    if lib.stochqn_hip_comm_unique_id(buf) != 0:
        pytest.fail("RCCL could not be loaded")
    assert lib.stochqn_hip_comm_init(0, 1, bytes(buf)) == 0
    try:
        assert lib.stochqn_hip_comm_nranks() == 1
        for cfg in [c for c in CONFIGS if c[0] in ("sqn_hessvec", "olbfgs_default", "adaqn_fisher_rms", "sqn_graddiff")]:
            got, want = both_traces(cfg, 1000, "device", hip_backend, oracle_backend)
            compare_traces(got, want, FREE_RUN_TOL.get(cfg[0], TOL))
    finally:
        lib.stochqn_hip_comm_finalize()


# ---------------------------------------------------------------------------------------------
# sharded path of the LIBRARY ITSELF on one GPU: P shards, P host threads, loop-back all-reduce
# ---------------------------------------------------------------------------------------------
def _sharded_run(optname, kw, P_full, nshards, step, calls, hip_backend):
    import threading
    import stochqn_amd
    lib = stochqn_amd.cdll()
    n = P_full.n
    bounds = [(n * r // nshards, n * (r + 1) // nshards) for r in range(nshards)]
    bounds[0] = (0, bounds[0][1] + 1) if nshards > 1 and bounds[0][1] + 1 < bounds[1][1] else bounds[0]   # uneven on purpose
    if nshards > 1:
        bounds[1] = (bounds[0][1], bounds[1][1])
    traces = [None] * nshards
    errors = []
    fparts = [0.0] * nshards
    rendezvous = threading.Barrier(nshards)
    assert lib.stochqn_hip_loopback_init(nshards) == 0

    def worker(r):
        try:
            assert lib.stochqn_hip_loopback_join(r) == 0
            lo, hi = bounds[r]
            opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
            x = P_full.x0()[lo:hi].copy()
            tr = []
            last = 999983
            for call in range(calls):
                res = opt.run_optimizer(x, step)
                task, req = res["task"], res["requested_on"]
                tr.append({"task": task, "info": res["info"]["iteration_info"], "niter": opt.niter, "section": opt.section,
                           "mem_used": opt.BFGS_mem.mem_used, "mem_st_ix": opt.BFGS_mem.mem_st_ix, "x": x.copy()})
                if task in ("calc_grad", "calc_grad_same_batch", "calc_grad_big_batch"):
                    if task == "calc_grad":
                        last = call
                    c = last if task == "calc_grad_same_batch" else call
                    full = np.zeros(n)
                    full[lo:hi] = req
                    opt.update_gradient(P_full.grad(full, c)[lo:hi])          # diagonal problem: g_i depends on x_i only
                elif task == "calc_hess_vec":
                    opt.update_hess_vec(P_full.d[lo:hi] * req[1])
                elif task == "calc_fun_val_batch":
                    fparts[r] = 0.5 * float(np.sum(P_full.d[lo:hi] * req * req))
                    rendezvous.wait()
                    f = sum(fparts) * (10.0 if call in P_full.f_spike_calls else 1.0)
                    rendezvous.wait()
                    opt.update_function(f)
            traces[r] = tr
        except Exception as e:                                   # pragma: no cover
            errors.append((r, repr(e)))
            try:
                rendezvous.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(nshards)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    alive = [t.is_alive() for t in threads]
    lib.stochqn_hip_loopback_finalize()
    assert not any(alive), "a shard thread hung"
    assert not errors, errors
    return traces, bounds


@pytest.mark.parametrize("nshards", [2, 3])
@pytest.mark.parametrize("name", ["sqn_hessvec", "olbfgs_default", "sqn_graddiff", "adaqn_fisher_rms", "sqn_nan", "olbfgs_reject_all", "adaqn_fisher500"])
def test_sharded_library_equals_unsharded_oracle(name, nshards, form, hip_backend, oracle_backend):
    """Every shard runs the real kernels on its slice; every reduction goes k_fin -> all-reduce
    (loop-back) -> consumer, exactly as with RCCL.  All shards must take the same decisions as the
    unsharded oracle and their slices must concatenate to its x."""
    cfg = [c for c in CONFIGS if c[0] == name][0]
    _, optname, kw, step, calls, pkw = cfg
    n = 3001
    P = NoisyQuadratic(n, seed=7, **pkw)
    want = run_trace(OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw), P, P.x0(), step, calls)
    traces, bounds = _sharded_run(optname, kw, P, nshards, step, calls, hip_backend)
    for i, w in enumerate(want):
        for r in range(nshards):
            g = traces[r][i]
            for k in ("task", "info", "niter", "section", "mem_used", "mem_st_ix"):
                assert g[k] == w[k], (i, r, k, g[k], w[k])
        x = np.concatenate([traces[r][i]["x"] for r in range(nshards)])
        assert rel_err(x, w["x"]) <= FREE_RUN_TOL.get(name, TOL), (i, rel_err(x, w["x"]))


def test_invalid_workspace_is_refused(hip_backend):
    """Error convention of reference src/stochqn.c:1033-1035,1144-1146,1293-1295: a section the state
    machine does not know -> task = invalid_input, return -1000, nothing touched."""
    from stochqn_amd import _abi
    torch = torch_cuda()
    n = 16
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    g = torch.ones(n, dtype=torch.float64, device="cuda")
    S = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
    Y = torch.zeros(3 * n, dtype=torch.float64, device="cuda")
    rho, alpha = np.zeros(3), np.zeros(3)
    b = _abi.bfgs_mem(S.data_ptr(), Y.data_ptr(), rho.ctypes.data, alpha.ctypes.data, None, None, 3, 0, 0, 1, 0.0, 0.0)
    gp = torch.zeros(n, dtype=torch.float64, device="cuda")
    w = _abi.workspace_oLBFGS(C.pointer(b), gp.data_ptr(), 0.0, 0, 7, 1, 1, n)          # section 7 does not exist
    req, task, info = C.c_void_p(), C.c_int(), C.c_int()
    rc = hip_backend.run_oLBFGS(0.1, x.data_ptr(), g.data_ptr(), C.byref(req), C.byref(task), C.byref(w), C.byref(info))
    assert rc == -1000 and task.value == 100 and info.value == 200
    assert torch.equal(x, torch.ones_like(x)) and w.niter == 0 and w.section == 7
    ws = _abi.workspace_SQN(C.pointer(b), None, gp.data_ptr(), gp.data_ptr(), 0, 0, -1, 1, 1, n)
    rv = C.c_void_p()
    rc = hip_backend.run_SQN(0.1, x.data_ptr(), g.data_ptr(), g.data_ptr(), C.byref(req), C.byref(rv), C.byref(task), C.byref(ws), C.byref(info))
    assert rc == -1000 and task.value == 100


def test_library_owned_workspaces_all_three(form, hip_backend, oracle_backend):
    """Profile A (initialize_* / run_* / dealloc_*) for each optimiser with a DEVICE caller: the
    arrays inside the returned structs are device memory, *req is a device pointer."""
    torch = torch_cuda()
    n = 513
    P = NoisyQuadratic(n, seed=4)

    def drive(be, make, run, free, device):
        w = make(be)
        x = torch.as_tensor(P.x0(), device="cuda") if device else P.x0()
        g = torch.zeros(n, dtype=torch.float64, device="cuda") if device else np.zeros(n)
        ptr = (lambda a: a.data_ptr()) if device else (lambda a: a.ctypes.data)
        req, task, info = C.c_void_p(), C.c_int(), C.c_int()
        xs = []
        for call in range(40):
            rc = run(be, w, ptr(x), ptr(g), req, task, info)
            assert rc in (0, 1)
            xs.append((rc, task.value, info.value, x.cpu().numpy().copy() if device else x.copy()))
            if device:
                view = torch.empty(0)      # *req is a device pointer: rebuild the argument on the host from x / workspace
                if req.value == x.data_ptr():
                    at = x.cpu().numpy()
                else:
                    buf = torch.zeros(n, dtype=torch.float64, device="cuda")
                    assert torch.cuda.current_stream().synchronize() is None
                    import ctypes
                    hip = ctypes.CDLL("libamdhip64.so")
                    assert hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(req.value), ctypes.c_size_t(8 * n), 3) == 0
                    at = buf.cpu().numpy()
            else:
                at = np.ctypeslib.as_array(C.cast(req.value, C.POINTER(C.c_double)), (n,)).copy()
            if task.value in (101, 102, 103):
                gv = P.grad(at, call)
                if device:
                    g.copy_(torch.as_tensor(gv))
                else:
                    g[:] = gv
        free(be, w)
        return xs

    cases = {
        "oLBFGS": (lambda be: be.initialize_oLBFGS(n, 4, 0.0, 0.0, 1e-4, 1, 1),
                   lambda be, w, x, g, req, task, info: be.run_oLBFGS(0.1, x, g, C.byref(req), C.byref(task), w, C.byref(info)),
                   lambda be, w: be.dealloc_oLBFGS(w)),
        "SQN-graddiff": (lambda be: be.initialize_SQN(n, 3, 4, 1e-4, 1, 0.0, 1, 1),
                         lambda be, w, x, g, req, task, info: be.run_SQN(0.1, x, g, None, C.byref(req), C.byref(C.c_void_p()), C.byref(task), w, C.byref(info)),
                         lambda be, w: be.dealloc_SQN(w)),
        "adaQN-graddiff": (lambda be: be.initialize_adaQN(n, 3, 5, 4, 0.0, 1e-4, 1e-4, 0.9, 1, 0.0, 1, 1),
                           lambda be, w, x, g, req, task, info: be.run_adaQN(0.05, x, 0.0, g, C.byref(req), C.byref(task), w, C.byref(info)),
                           lambda be, w: be.dealloc_adaQN(w)),
    }
    for name, (make, run, free) in cases.items():
        want = drive(oracle_backend, make, run, free, device=False)
        got = drive(hip_backend, make, run, free, device=True)
        for i, (gg, ww) in enumerate(zip(got, want)):
            assert gg[:3] == ww[:3], (name, i, gg[:3], ww[:3])
            assert rel_err(gg[3], ww[3]) <= TOL, (name, i)


# ---------------------------------------------------------------------------------------------
# single-precision ABI (libstochqn_f32.so; reference -DUSE_FLOAT build)
# ---------------------------------------------------------------------------------------------
F32_TOL = 2e-4      # vectors are stored as float (6e-8 per rounding); the oracle rounds after every axpy,
                    # the GPU accumulates in double and rounds once per sweep / pass
F32_CONFIGS = ["olbfgs_default", "olbfgs_nocurv_hess_init", "sqn_hessvec", "sqn_graddiff", "sqn_nan", "sqn_ring20",
               "adaqn_fisher_rms", "adaqn_graddiff_nomaxincr", "adaqn_nonan_check"]


@pytest.fixture(scope="session")
def hip_backend_f32():
    import stochqn_amd
    be = stochqn_amd.lib(use_float=True)
    assert stochqn_amd.cdll(use_float=True).stochqn_hip_available() == 1
    return be


@pytest.fixture(params=["threepass", "sweeps"])
def form_f32(request, hip_backend_f32):
    import stochqn_amd
    if request.param == "sweeps" and getattr(request.node, "callspec", None) is not None \
            and request.node.callspec.params.get("n") in SWEEPS_SKIP_N:
        pytest.skip("the fallback form runs the ends of each size grid, the default form all of it")
    lib = stochqn_amd.cdll(use_float=True)
    set_form(lib, request.param)
    yield request.param
    reset_form(lib)


@pytest.mark.parametrize("n", [5, 64, 1000, 4099])
@pytest.mark.parametrize("name", F32_CONFIGS)
def test_float_lockstep_parity(name, n, form_f32, hip_backend_f32):
    """Same lock-step protocol as the fp64 test, float arrays on both sides (oracle built -DUSE_FLOAT)."""
    import stochqn_amd
    from oracle import oracle
    cfg = [c for c in CONFIGS if c[0] == name][0]
    _, optname, kw, step, calls, pkw = cfg
    P = NoisyQuadratic(n, seed=11, **pkw)
    ref = OPTIMIZERS[optname](backend=oracle.bound_f32(), space="host", use_float=True, **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend_f32, space="device", use_float=True, **kw)
    x_ref = P.x0().astype(np.float32)
    x_dev = torch_cuda().as_tensor(x_ref.copy(), device="cuda")
    lib = stochqn_amd.cdll(use_float=True)
    inval = lambda o: lib.stochqn_hip_invalidate(C.c_void_p(o._sp.ptr(o.BFGS_mem.s_mem)))
    run_lockstep(ref, opt, P, x_ref, x_dev, step, min(calls, 50), F32_TOL, on_sync=inval)


@pytest.mark.parametrize("name", ["adaqn_ring20"])
def test_float_lockstep_parity_full_grids(name, hip_backend_f32):
    """Single precision at full launch shapes: n = 1,000,003 puts three ring rows in four off the 16-byte grid
    (float4 packs read at 4-byte alignment), the row-split pass A runs its whole-rounds grid.  The one hard configuration
    (rounds 2 - 4 also ran sqn_ring20 and adaqn_fisher_rms here: the same kernels at the same launch shapes, 17 s of the
    suite's budget; in single precision they stay covered at every size of test_float_lockstep_parity's grid).

    adaqn_ring20 (L = 1, step 0.002) is the hard case: s = x_new - x_old cancels to ~1e-3 of x, so the ONE float rounding by
    which the library's update (fma in double, one rounding to float) and the oracle's (float product, float sum: two
    roundings, as -- per BLAS -- the reference's saxpy, src/stochqn.c:838) may differ in x shows up ~1000 times larger in s,
    and y = F'(F s)/fu, which cancels further, carries it on (3.1e-4 at call 7: round 2 dropped the case and blamed the float
    accumulation of y; the accumulation is double on both sides).  So those two rows are judged on what each side actually
    computes: the library's y against F'(F s)/fu evaluated from the library's OWN s (t rounded to float in between, as the
    reference's buffer_y does), at the common tolerance -- a yardstick that is first required to reproduce the oracle's y from
    the oracle's s; and s against the oracle's within the bound the cancellation allows, eps_float |x| / |s|, asserted."""
    import stochqn_amd
    from oracle import oracle
    _, optname, kw, step, calls, pkw = [c for c in CONFIGS if c[0] == name][0]
    n = 1_000_003
    P = NoisyQuadratic(n, seed=11, **pkw)
    ref = OPTIMIZERS[optname](backend=oracle.bound_f32(), space="host", use_float=True, **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend_f32, space="device", use_float=True, **kw)
    x_ref = P.x0().astype(np.float32)
    x_dev = torch_cuda().as_tensor(x_ref.copy(), device="cuda")
    lib = stochqn_amd.cdll(use_float=True)
    inval = lambda o: lib.stochqn_hip_invalidate(C.c_void_p(o._sp.ptr(o.BFGS_mem.s_mem)))
    judged = {"s": 0, "y": 0}

    def fisher_rows(where, rname, row, got, want, A_o, A_r):
        if name != "adaqn_ring20" or rname not in ("BFGS_mem.s_mem", "BFGS_mem.y_mem") or not np.any(want):
            return False
        eps = float(np.finfo(np.float32).eps)
        s_lib = np.asarray(to_np(A_o["BFGS_mem.s_mem"][row * n:(row + 1) * n]), dtype=np.float64)
        s_ref = np.asarray(A_r["BFGS_mem.s_mem"][row * n:(row + 1) * n], dtype=np.float64)
        amplification = float(np.linalg.norm(x_ref.astype(np.float64)) / np.linalg.norm(s_ref))
        if rname == "BFGS_mem.s_mem":
            assert rel_err(got, want) <= max(F32_TOL, eps * amplification), where
            judged["s"] += 1
            return True
        fu = ref.Fisher_mem.mem_used
        F = np.asarray(A_r["Fisher_mem.F"], dtype=np.float64)[:fu * n].reshape(fu, n)
        if rel_err(got, want) <= F32_TOL:
            return True                                              # an old pair, or a new one that did not cancel
        def product(s_):                                             # buffer_y is real_t: t is rounded on its way (reference :946-949)
            t = (F @ s_).astype(np.float32).astype(np.float64)
            return (F.T @ t) / fu
        assert rel_err(want, product(s_ref)) <= F32_TOL, where       # the yardstick reproduces the oracle on the oracle's s ...
        e_own = rel_err(got, product(s_lib))                         # ... so it may judge the library on the library's s
        assert e_own <= F32_TOL, "%s: y against F'(F s)/fu of the library's own s: %.3e" % (where, e_own)
        judged["y"] += 1
        return True
    run_lockstep(ref, opt, P, x_ref, x_dev, step, min(calls, 18), F32_TOL, on_sync=inval, row_check=fisher_rows)
    if name == "adaqn_ring20":
        assert judged["s"] >= 10 and judged["y"] >= 1                # the hook really saw the pairs, and the hard one


@pytest.mark.parametrize("space", ["host", "device"])
def test_float_known_answer_trajectory(space, form_f32, hip_backend_f32):
    """The reference's 2-D Rosenbrock SQN run (tests/golden/known_answers.json) in single precision:
    same iteration / request counts, x within float accuracy of the double answer."""
    k = GOLD["SQN_rosen2d"]
    from harness import Rosenbrock2D
    opt = OPTIMIZERS["SQN"](backend=hip_backend_f32, space=space, use_float=True)
    P = Rosenbrock2D()
    x = P.x0().astype(np.float32)
    if space == "device":
        x = torch_cuda().as_tensor(x, device="cuda")
    tr = run_trace(opt, P, x, k["step"], k["calls"])
    assert tr[-1]["niter"] == k["niter"] and tr[-1]["mem_used"] == k["mem_used"]
    assert sum(r["task"] == "calc_hess_vec" for r in tr) == k["n_hess_vec"]
    assert np.allclose(tr[-1]["x"], k["x"], rtol=2e-3)


# ---------------------------------------------------------------------------------------------
# randomised hyper-parameters (deterministic seeds): lock-step parity on configurations nobody hand-picked
# ---------------------------------------------------------------------------------------------
def _random_config(optname, rng):
    kw = dict(mem_size=int(rng.integers(1, 9)), check_nan=bool(rng.integers(0, 2)),
              min_curvature=[None, 1e-6, 1e-2][rng.integers(0, 3)], y_reg=[None, 1e-3][rng.integers(0, 2)])
    if optname == "oLBFGS":
        kw["hess_init"] = [None, 0.3][rng.integers(0, 2)]
        step = 0.1
    elif optname == "SQN":
        kw["bfgs_upd_freq"] = int(rng.integers(1, 6))
        kw["use_grad_diff"] = bool(rng.integers(0, 2))
        step = 0.1
    else:
        kw["bfgs_upd_freq"] = int(rng.integers(1, 6))
        kw["use_grad_diff"] = bool(rng.integers(0, 2))
        kw["fisher_size"] = int(rng.integers(1, 12))
        kw["max_incr"] = [None, 1.01, 1.5][rng.integers(0, 3)]
        kw["rmsprop_weight"] = [None, 0.9, 0.5][rng.integers(0, 3)]
        kw["scal_reg"] = [1e-4, 1e-2][rng.integers(0, 2)]
        step = 0.02
    return kw, step


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("optname", ["oLBFGS", "SQN", "adaQN"])
def test_lockstep_random_hyperparameters(optname, seed, form, hip_backend, oracle_backend):
    import stochqn_amd
    rng = np.random.default_rng([seed, len(optname)])
    kw, step = _random_config(optname, rng)
    n = int(rng.choice([6, 31, 256, 1025, 5000]))
    nan_calls = (int(rng.integers(5, 30)),) if rng.integers(0, 4) == 0 else ()
    P = NoisyQuadratic(n, seed=100 + seed, nan_calls=nan_calls)
    ref = OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
    x_ref = P.x0()
    x_dev = torch_cuda().as_tensor(P.x0(), device="cuda")
    lib = stochqn_amd.cdll()
    inval = lambda o: lib.stochqn_hip_invalidate(C.c_void_p(o._sp.ptr(o.BFGS_mem.s_mem)))
    run_lockstep(ref, opt, P, x_ref, x_dev, step, 36, TOL, on_sync=inval)


@pytest.mark.parametrize("n", [2147483646, 2147483647])
def test_two_loop_at_the_int_limit(n, form, hip_backend):
    """n is an `int` in the ABI (reference include/stochqn.h:172-174): the largest even and the largest
    odd n (16-byte packs vs single elements, 32-bit pack indices at their limit).  Secant equation:
    the two-loop maps the newest y onto the newest s."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    m = 2
    gen = torch.Generator(device="cuda").manual_seed(7)
    S = torch.empty(m * n, dtype=torch.float64, device="cuda")
    Y = torch.empty(m * n, dtype=torch.float64, device="cuda")
    chunk = 1 << 28
    for k in range(m):
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            s = 1e-3 * (torch.rand(hi - lo, dtype=torch.float64, device="cuda", generator=gen) - 0.5)
            d = 0.5 + torch.rand(hi - lo, dtype=torch.float64, device="cuda", generator=gen)
            S[k * n + lo:k * n + hi] = s
            Y[k * n + lo:k * n + hi] = d * s
            del s, d
    q = Y[n:2 * n].clone()                      # newest pair = row 1 (st = 0, used = 2)
    hip_two_loop(lib, q, None, 0.0, Y, S, n, m, m, 0)
    num = 0.0
    den = 0.0
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        s_new = S[n + lo:n + hi]
        num += float(torch.sum((q[lo:hi] - s_new) ** 2))
        den += float(torch.sum(s_new ** 2))
    lib.stochqn_hip_release(C.c_void_p(S.data_ptr()))
    assert (num / den) ** 0.5 <= TOL


# ---------------------------------------------------------------------------------------------
# out-of-bounds writes: every caller array sits between guard zones that must survive
# ---------------------------------------------------------------------------------------------
class _GuardedSpace:
    """Device arrays carved out of larger allocations with sentinel-filled guards on both sides.  The
    payload keeps the alignment class of a plain allocation when `shift` is 0 and is deliberately
    misaligned (8-byte only) when `shift` is 1."""
    GUARD = 64
    SENTINEL = -777.25

    def __init__(self, base_space, shift):
        self._b = base_space
        self.torch = base_space.torch
        self.device = base_space.device
        self.dtype = base_space.dtype
        self.tdtype = base_space.tdtype
        self.name = "device"
        self.shift = shift
        self.bases = []

    def empty(self, n):
        n = int(n)
        base = self.torch.full((n + 2 * self.GUARD + 2,), self.SENTINEL, dtype=self.tdtype, device=self.device)
        lo = self.GUARD + self.shift
        base[lo:lo + n] = 0
        self.bases.append((base, lo, n))
        return base[lo:lo + n]

    zeros = empty

    def ptr(self, a):
        return a.data_ptr()

    def assign(self, dst, src):
        self._b.assign(dst, src)

    def is_array(self, a):
        return self._b.is_array(a)

    def __getattr__(self, name):                      # whatever else the package asks of a space (pin, unpin_all, attach ...)
        return getattr(self._b, name)

    def check(self):
        for base, lo, n in self.bases:
            assert bool((base[:lo] == self.SENTINEL).all()), "write below an array"
            assert bool((base[lo + n:] == self.SENTINEL).all()), "write beyond an array"


@pytest.mark.parametrize("shift", [0, 1])
@pytest.mark.parametrize("n", [1, 127, 128, 129, 1000, 4099])
@pytest.mark.parametrize("name", ["olbfgs_default", "sqn_hessvec", "sqn_graddiff", "adaqn_fisher_rms", "adaqn_graddiff", "sqn_ring20"])
def test_no_out_of_bounds_writes(name, n, shift, form, hip_backend):
    cfg = [c for c in CONFIGS if c[0] == name][0]
    _, optname, kw, step, calls, pkw = cfg
    P = NoisyQuadratic(n, seed=3, **pkw)
    opt = OPTIMIZERS[optname](backend=hip_backend, space="device", **kw)
    guarded = _GuardedSpace(opt._sp, shift)
    opt._sp = guarded
    x = guarded.empty(n)
    x.copy_(torch_cuda().as_tensor(P.x0()))
    run_trace(opt, P, x, step, min(calls, 45))
    torch_cuda().cuda.synchronize()
    guarded.check()
    assert bool(torch_cuda().isfinite(x).all()) or "nan" in name


# ---------------------------------------------------------------------------------------------
# allocation failures (reference src/stochqn.c:479,504,545: print, clean up, report -- never crash)
# ---------------------------------------------------------------------------------------------
class _RetryOnce:
    """Wraps an optimiser: a call refused with "invalid workspace" while an allocation failure is
    injected is repeated once with the injection off.  A refused call must not have touched x, the
    counters or the device state, so the trace has to come out as if nothing had happened."""

    def __init__(self, opt, setopt):
        self.__dict__["_opt"], self.__dict__["_setopt"], self.__dict__["refused"] = opt, setopt, 0

    def __getattr__(self, name):
        return getattr(self._opt, name)

    def __setattr__(self, name, value):
        setattr(self._opt, name, value)

    def run_optimizer(self, x, step):
        try:
            return self._opt.run_optimizer(x, step)
        except ValueError:
            self._setopt(b"fail_alloc_after", -1.0)
            self.__dict__["refused"] += 1
            return self._opt.run_optimizer(x, step)


@pytest.mark.parametrize("space", ["host", "device"])
@pytest.mark.parametrize("cfgname", ["olbfgs_default", "sqn_hessvec", "sqn_graddiff", "adaqn_fisher_rms"])
def test_allocation_failures_are_refused_cleanly(cfgname, space, hip_backend, oracle_backend, capfd):
    """Fault injection (option fail_alloc_after = k: the (k+1)-th device / pinned allocation from now
    fails once).  For every k until the run needs no more allocations: the call that hits the
    failure returns -1000 with task = invalid_input and a message on stderr, leaves the optimiser
    untouched, and the same call repeated afterwards succeeds -- the whole trace still equals the
    oracle's."""
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    cfg = {c[0]: c for c in CONFIGS}[cfgname]
    _, optname, kw, step, calls, pkw = cfg
    n = 300
    P = NoisyQuadratic(n, seed=7, **pkw)
    want = run_trace(OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw), P, P.x0(), step, calls)
    refused_total = 0
    try:
        for k in range(40):
            opt = _RetryOnce(OPTIMIZERS[optname](backend=hip_backend, space=space, **kw), lib.stochqn_hip_set_option)
            x = P.x0() if space == "host" else torch_cuda().as_tensor(P.x0(), device="cuda")
            lib.stochqn_hip_set_option(b"fail_alloc_after", float(k))
            got = run_trace(opt, P, x, step, calls)
            lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
            compare_traces(got, want, FREE_RUN_TOL.get(cfgname, TOL))
            refused_total += opt.refused
            opt._opt.release()
            if opt.refused == 0:          # k allocations were not even reached: nothing left to break
                break
        else:
            raise AssertionError("more than 40 allocations in one short run")
    finally:
        lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
    assert refused_total >= 2            # at least the scratch pool and the pinned read-back block
    assert "could not allocate" in capfd.readouterr().err


def test_initialize_reports_allocation_failure(hip_backend, capfd):
    """initialize_* with an allocation that fails -> NULL + message, nothing leaked that a later
    call would trip over (reference src/stochqn.c:479,504,545)."""
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    lib.initialize_SQN.restype = C.c_void_p
    lib.initialize_SQN.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]
    lib.initialize_adaQN.restype = C.c_void_p
    lib.initialize_adaQN.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]
    lib.dealloc_SQN.argtypes = [C.c_void_p]
    lib.dealloc_adaQN.argtypes = [C.c_void_p]
    try:
        for k in range(6):
            lib.stochqn_hip_set_option(b"fail_alloc_after", float(k))
            assert lib.initialize_SQN(1000, 5, 10, 1e-4, 0, 0.0, 1, 1) is None
        for k in range(9):
            lib.stochqn_hip_set_option(b"fail_alloc_after", float(k))
            assert lib.initialize_adaQN(1000, 5, 16, 10, 1.01, 1e-4, 1e-4, 0.9, 0, 0.0, 1, 1) is None
    finally:
        lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
    assert "Could not allocate memory" in capfd.readouterr().err
    # far beyond the 288 GB of the device: the real thing, not the injected one
    assert lib.initialize_SQN(2**31 - 1, 64, 10, 0.0, 0, 0.0, 1, 1) is None
    w = lib.initialize_SQN(1000, 5, 10, 1e-4, 0, 0.0, 1, 1)
    assert w is not None
    lib.dealloc_SQN(w)


def test_library_loaded_before_torch_still_takes_torch_tensors():
    """Import order must not matter (stochqn_amd._share_hip_runtime_with_torch): a fresh process loads
    libstochqn.so first, runs a host-caller optimisation, then imports torch and hands the library
    torch device tensors."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, stochqn_amd\n"
        "from harness import NoisyQuadratic, run_trace, OPTIMIZERS\n"
        "be = stochqn_amd.lib(); P = NoisyQuadratic(300, seed=7)\n"
        "h = run_trace(OPTIMIZERS['SQN'](backend=be, space='host', mem_size=3, bfgs_upd_freq=4), P, P.x0(), 0.1, 40)\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "x = torch.as_tensor(P.x0(), device='cuda')\n"
        "d = run_trace(OPTIMIZERS['SQN'](backend=be, space='device', mem_size=3, bfgs_upd_freq=4), P, x, 0.1, 40)\n"
        "assert np.array_equal(h[-1]['x'], d[-1]['x']) and h[-1]['niter'] == d[-1]['niter']\n"
        "print('order-ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "order-ok" in out.stdout, out.stdout + out.stderr


def test_concurrent_workspaces_from_two_threads(form, hip_backend, oracle_backend):
    """Threading contract of SURVEY.md 8b: no global optimiser state, re-entrant across DISTINCT
    workspaces.  Three host threads drive three different optimisers at the same time (ctypes drops the
    GIL inside run_*; every context has its own stream); each trace must equal the oracle's."""
    import threading
    torch = torch_cuda()
    by_name = {c[0]: c for c in CONFIGS}
    picks = [("sqn_hessvec", "device"), ("adaqn_fisher_rms", "host"), ("olbfgs_default", "device")]
    n = 2000
    wants, results, errors = {}, {}, []
    for name, _ in picks:
        _, optname, kw, step, calls, pkw = by_name[name]
        P = NoisyQuadratic(n, seed=7, **pkw)
        wants[name] = run_trace(OPTIMIZERS[optname](backend=oracle_backend, space="host", **kw), P, P.x0(), step, calls)

    def work(name, space):
        try:
            _, optname, kw, step, calls, pkw = by_name[name]
            P = NoisyQuadratic(n, seed=7, **pkw)
            for rep in range(3):                       # several optimiser objects per thread, back to back
                opt = OPTIMIZERS[optname](backend=hip_backend, space=space, **kw)
                x = P.x0() if space == "host" else torch.as_tensor(P.x0(), device="cuda")
                results[(name, rep)] = run_trace(opt, P, x, step, calls)
                opt.release()
        except Exception as e:                         # surfaced in the main thread
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=work, args=p) for p in picks]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for (name, rep), got in results.items():
        compare_traces(got, wants[name], FREE_RUN_TOL.get(name, TOL))
    assert len(results) == 9


def test_contexts_do_not_leak_device_memory(hip_backend):
    """R and Python never call dealloc_*: contexts are dropped by release(), by a new optimiser at the same
    address (section 0) or by stochqn_hip_release_all().  Two hundred host-caller optimisers (mirrors of
    S, Y, Fisher ring, staging and pinned buffers each) must leave the device's free memory where it was."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    n = 50_000
    P = NoisyQuadratic(n, seed=3)

    def cycle(release_how):
        for optname, kw in (("SQN", dict(mem_size=8, bfgs_upd_freq=3)),
                            ("adaQN", dict(mem_size=6, fisher_size=12, bfgs_upd_freq=3, max_incr=None)),
                            ("oLBFGS", dict(mem_size=8))):
            opt = OPTIMIZERS[optname](backend=hip_backend, space="host", **kw)
            run_trace(opt, P, P.x0(), 0.05, 12)
            if release_how == "release":
                opt.release()
            # else: left to __del__ / release_all

    cycle("release")                                 # warm the allocator caches of the runtime
    lib.stochqn_hip_release_all()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for rep in range(70):
        cycle("release" if rep % 2 else "gc")
    lib.stochqn_hip_release_all()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    per_object = 2 * 8 * n * 8                       # roughly the S/Y mirrors of one optimiser
    assert free0 - free1 < per_object, "device memory shrank by %d bytes over 210 optimiser objects" % (free0 - free1)


def test_bench_multi_process_control_flow_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), rehearsed
    with 3 ranks on the one GPU of the test box: `--rehearse` puts every rank on cuda:0, runs
    torch.distributed over gloo and feeds the library's reductions through stochqn_hip_comm_init_custom.
    Everything else -- sharded data, per-rank contexts, identical decisions on all ranks, barriers, MAX over
    ranks, the untimed reference-form steps, one JSON line from rank 0 -- is the code of the real N-GPU run."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "3", "--rehearse", "--vars-per-gpu", "3000001",
           "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--sustain-seconds", "0.5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["steps"] == 12 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["f_end"] < d["config"]["f_start"]
    assert d["config"]["hess_vec_requests"] >= 1 and d["config"]["rejected_steps"] == 0
    assert d["reference_form"] is not None and d["cpu_baseline"] is None
    assert d["sustained"]["steps"] >= 10                            # every rank derived the same number of extra steps
    assert "REHEARSAL" in d["config"]["parallelism"]


def _bench(args, timeout=600, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in (env or os.environ).items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if "--sustain-seconds" not in args and "--in-process" not in args:
        args = args + ["--sustain-seconds", "0"]               # the tests that want the sustained leg ask for it
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                         cwd=root, env=env)
    return out


def test_bench_starts_its_own_ranks_and_shards_one_problem(tmp_path):
    """`python bench.py --gpus 3` with NO launcher: the parent decides before touching torch / HIP, starts 3 fresh
    ranks (torch.distributed.run child) and relays rank 0's single JSON line (rehearsed on the one GPU of the
    box, reductions over gloo).  The inputs come from the counter-based generator, so the 3-rank problem IS the
    1-rank problem: the concatenated x of the ranks must equal the x of a 1-rank run over n = 3 x 3,000,001."""
    per = 3_000_001
    common = ["--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-host-caller", "--no-live-pmc", "--no-reference-form"]
    out = _bench(["--gpus", "3", "--rehearse", "--vars-per-gpu", str(per), "--dump-x", str(tmp_path / "x3")] + common)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and len(d["per_rank_ms_per_step"]) == 3
    assert d["rccl_nranks"] == 0 and d["degraded"] is True          # a rehearsal: the reductions went over gloo, and the line says so
    assert d["steps"] == 12 and d["value"] > 0 and d["config"]["hess_vec_requests"] >= 1
    one = _bench(["--gpus", "1", "--vars-per-gpu", str(3 * per), "--dump-x", str(tmp_path / "x1")] + common)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["rccl_nranks"] == 1
    x3 = np.concatenate([np.load(str(tmp_path / ("x3.%d.npy" % r))) for r in range(3)])
    x1 = np.load(str(tmp_path / "x1.0.npy"))
    assert x3.shape == x1.shape
    assert rel_err(x3, x1) <= TOL, rel_err(x3, x1)
    assert abs(d["config"]["f_end"] - d1["config"]["f_end"]) <= 1e-9 * abs(d1["config"]["f_end"])
    assert d["config"]["calls"] == d1["config"]["calls"]


def test_bench_default_multi_gpu_run_carries_every_leg():
    """`python bench.py --gpus 3 --rehearse` and NOTHING else -- the shape of the driver's command for N > 1: after the
    primary weak-scaling leg the same JSON line must carry BASELINE config 5's weak-scaling point (`c5`), the strong-scaling
    split of config 3 (`strong`), the latency of one reduction on the library's communicator with the number of reductions a
    step issues (`allreduce_us`) and the one-process / N-devices mode as a fresh child process (`in_process`).  Rehearsed
    with 3 ranks on the one GPU (every n divided by 50, reductions over gloo): the control flow is the N-GPU run's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rehearse"], capture_output=True, text=True,
                         timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["scaling"] == "weak" and d["config"]["name"] == "c3"
    # a rehearsal's reductions travel over gloo: the line says at the top level that RCCL did not produce this number
    assert d["rccl_nranks"] == 0 and d["degraded"] is True and "RCCL" in d["degraded_because"][0]
    assert d["legs_failed"] == [], d["legs"]
    legs = d["legs"]
    assert set(legs) == {"c5", "strong", "allreduce_us", "in_process"}
    assert legs["c5"]["n_per_gpu"] == 125_000_000 // 50 and legs["c5"]["steps"] >= 20 and legs["c5"]["objective_fell"]
    assert legs["c5"]["hess_vec_requests"] >= 2 and legs["c5"]["rejected_steps"] == 0 and legs["c5"]["steps_per_s"] > 0
    # config 5's yardstick -- one GPU at the same per-GPU shard -- was measured on this very node, after the ranks had gone
    ref = legs["c5"]["shard_reference_1gpu"]
    assert ref["source"] == "this node" and ref["measured"]["n_per_gpu"] == 125_000_000 // 50 and ref["steps_per_s"] > 0
    assert legs["c5"]["this_run_over_reference"] > 0 and isinstance(legs["c5"]["within_15pct_of_linear"], bool)
    assert legs["strong"]["n_total"] == 3 * (100_000_000 // 50 // 3) and legs["strong"]["objective_fell"]
    assert legs["allreduce_us"]["median_us"] > 0 and legs["allreduce_us"]["min_us"] <= legs["allreduce_us"]["median_us"]
    assert 3.0 <= legs["allreduce_us"]["allreduces_per_step"] <= 4.5        # three per step in the three-pass form + the pair work every L
    assert abs(legs["c5"]["allreduces_per_step"] - legs["allreduce_us"]["allreduces_per_step"]) < 0.5
    ip = legs["in_process"]
    assert ip["device_shards"] == 3 and ip["n_per_gpu"] == 100_000_000 // 50 and ip["steps_per_s"] > 0 and ip["rejected_steps"] == 0
    assert 3.0 <= ip["allreduces_per_step"] <= 4.5 and ip["allreduce_us"]["median_us"] > 0
    assert d["forms"]["three_pass"] == d["steps"] and d["forms"]["sweeps"] == 0


def test_bench_falls_back_to_the_host_reducer_when_rccl_cannot_be_set_up():
    """The harness's own collectives go over gloo; only the library's reductions need RCCL.  If that communicator cannot be
    brought up (test hook: the init is declared failed on a one-rank group), the run goes on with the reductions over gloo and
    says so in the line instead of dying without one."""
    env = dict(os.environ, BENCH_TEST_RCCL_FAILS="1")
    out = _bench(["--force-dist", "--vars-per-gpu", "3000000", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-host-caller",
                  "--no-live-pmc", "--sustain-seconds", "0"], timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d["reducer"].startswith("gloo (the library") and d["rccl_nranks"] == 0 and "COULD NOT BE BROUGHT UP" in d["config"]["parallelism"]
    assert d["degraded"] is True
    assert d["value"] > 0 and d["allreduces_per_step"] >= 3
    assert "could not be set up" in out.stderr
    ok = _bench(["--force-dist", "--vars-per-gpu", "3000000", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-host-caller",
                 "--no-live-pmc", "--sustain-seconds", "0"], timeout=600)
    d2 = json.loads([l for l in ok.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d2["reducer"] == "rccl" and d2["rccl_nranks"] == 1 and d2["degraded"] is False
    # ... and when the communicator's set-up never RETURNS (ncclCommInitRank is a rendezvous: a peer or a fabric that does not answer
    # at first contact): the init runs on a thread of its own under BENCH_RCCL_INIT_S, the ranks agree, the run goes over gloo
    hung = _bench(["--force-dist", "--vars-per-gpu", "3000000", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-host-caller",
                   "--no-live-pmc", "--sustain-seconds", "0"], timeout=600, env=dict(os.environ, BENCH_TEST_RCCL_INIT_HANGS="1", BENCH_RCCL_INIT_S="3"))
    assert hung.returncode == 0, hung.stderr[-3000:]
    d3 = json.loads([l for l in hung.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d3["reducer"].startswith("gloo (the library") and d3["rccl_nranks"] == 0 and d3["degraded"] is True and d3["value"] > 0
    assert "had not returned after 3 s" in hung.stderr


def test_bench_keeps_its_primary_result_when_an_auxiliary_leg_hangs():
    """First contact with RCCL on N > 1 ranks happens on the driver's node: a collective of an auxiliary leg that never
    completes must not cost the run its primary (weak-scaling) number.  With the `c5` leg made to hang (test hook) and the
    watchdog's limit at 5 s, the ONE line still comes, with the primary result and the leg named in `legs_failed`; every rank
    leaves with code 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_HANG_LEG="c5", BENCH_WATCHDOG_S="5")
    # (the control flow is what is under test: a short primary leg -- the default run's 200 steps, sustained seconds, repeated
    # regions and reference-form extras are test_bench_default_multi_gpu_run_carries_every_leg's business)
    short = ["--steps", "20", "--warmup", "2", "--sustain-seconds", "0", "--value-runs", "1", "--no-reference-form", "--no-profile"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rehearse"] + short, capture_output=True, text=True,
                         timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["value"] > 0 and d["config"]["name"] == "c3"
    assert d["legs"] == {} and len(d["legs_failed"]) == 1 and "watchdog" in d["legs_failed"][0] and "'c5'" in d["legs_failed"][0]
    assert d["degraded"] is True and any("watchdog" in w for w in d["degraded_because"])
    # --strict-legs: the same hang is a failed run -- no line, a non-zero exit code from every rank
    strict = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rehearse", "--strict-legs"] + short, capture_output=True,
                            text=True, timeout=600, cwd=root, env=env)
    assert strict.returncode != 0 and not [l for l in strict.stdout.splitlines() if l.startswith('{"metric"')], strict.stdout[-1000:]


def test_bench_prints_its_one_line_inside_its_budget_and_names_what_it_left_out():
    """ONE wall-clock budget for the whole command (BENCH_BUDGET_S, started at process entry and handed on to every
    child process): an auxiliary leg runs only if its measured cost still fits, and whatever happens the line is printed no later
    than the budget.  (a) 45 s for the 3-rank rehearsal of the driver's N > 1 command: the in-process child (cost on file: 40 s
    ... at full size, more than what is left) is skipped and NAMED, every leg that fits still runs, one line, on time.
    (b) a leg that hangs while the per-leg watchdog is far away (300 s): the budget itself cuts the run off -- the line comes at the
    deadline with the primary result, `degraded`, and the leg that was running named in `legs_skipped`."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_DEADLINE_EPOCH")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rehearse"], capture_output=True, text=True,
                         timeout=600, cwd=root, env=dict(env, BENCH_BUDGET_S="45"))
    took = time.time() - t0
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert took < 45 + 15, took
    assert d["n_gpus"] == 3 and d["value"] > 0 and d["roofline"] is not None
    assert "in_process" in d["legs_skipped"] and d["legs_failed"] == []
    sk = {s["leg"]: s for s in d["budget"]["skipped"]}
    assert sk["in_process"]["needs_s"] > sk["in_process"]["left_s"] and d["budget"]["deadline_inherited"] is True
    assert {"c5", "strong", "allreduce_us"} <= set(d["legs"]) and "in_process" not in d["legs"]      # what fits still runs
    assert d["budget"]["leg_seconds"]["c5"] > 0 and d["budget"]["used_s"] <= 45
    # (b) the budget runs out under a leg that never comes back
    short = ["--steps", "20", "--warmup", "2", "--sustain-seconds", "0", "--value-runs", "1", "--no-reference-form"]
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rehearse"] + short, capture_output=True, text=True,
                         timeout=600, cwd=root, env=dict(env, BENCH_BUDGET_S="30", BENCH_TEST_HANG_LEG="strong", BENCH_WATCHDOG_S="300"))
    took = time.time() - t0
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert 20 < took < 30 + 15, took
    assert d["value"] > 0 and d["degraded"] is True and any("budget" in w for w in d["degraded_because"])
    assert "strong" in d["legs_skipped"] and "c5" in d["legs"] and "strong" not in d["legs"]


def _host_memory_available():
    avail = 0
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            avail = int(line.split()[1]) * 1024
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            avail = min(avail, int(lim) - int(open("/sys/fs/cgroup/memory.current").read()))
    except (OSError, ValueError):
        pass
    return avail


def test_the_benchmarked_workload_itself_matches_the_oracle(hip_backend, oracle_backend):
    """bench.py's headline workload EXACTLY as it is timed -- `bench.Workload`: SQN, n = 1e8, m = 20, L = 10, the ring pre-filled
    and wrapped (mem_st_ix = 3), Hv = A'(Av)/32 over the 32-row mini-batch (stochqn_hip_fisher_product), check_nan = 1, the
    default three-pass form -- against the CPU oracle from the same state: one whole L-cycle (ten ordinary steps), the
    Hessian-vector request at x_avg that builds a pair from the batch (reference src/stochqn.c:1093-1115, 962-966), and the
    step after the pair, which is the first to use it.  Identical task / info / counter sequences; x, the averages, the NEW
    row of S and of Y, rho and alpha of the last step at 1e-10.  Both sides get their gradient from their own iterate by
    the same expression ((d x) noise_t, noise_t from the counter-based generator: bit-identical in numpy and torch), so only
    the library's arithmetic can differ.  (the full-size SQN parity test uses Hv = d v at L = 1.)"""
    import bench
    import stochqn_amd
    from oracle import oracle
    from stochqn_amd import _abi
    torch = torch_cuda()
    n, m, L, bs = 100_000_000, 20, 10, 32
    need = (2 * m + bs + 9) * n * 8
    avail = _host_memory_available()
    if avail < 1.2 * need:
        pytest.skip("the host has %.0f GB available, the oracle's copy of the workload needs %.0f GB" % (avail / 1e9, 1.2 * need / 1e9))
    lib = stochqn_amd.cdll()
    bench.prototypes(lib)
    ctx = {"lib": lib, "be": hip_backend, "dev": torch.device("cuda", 0), "dist": None, "cpu_or_dev": "cpu", "rank": 0, "world": 1}
    wl = bench.Workload(ctx, n, 0, m, L, bs)
    olib = oracle.cdll()
    oracle.set_threads(oracle.usable_cpus())
    try:
        assert (wl.b.mem_used, wl.b.mem_st_ix, wl.w.niter, wl.w.section, wl.w.check_nan, wl.b.upd_freq) == (m, 3, L, 1, 1, L)
        # the oracle's copy of the state the timed region starts from
        def rows_to_host(t, rows):                                     # device -> the numpy buffer itself, row by row (as the 128-row Fisher test does)
            out = np.empty(rows * n)
            if hasattr(olib, "oracle_first_touch"):
                olib.oracle_first_touch(out.ctypes.data, rows * n)
            for k in range(rows):
                torch.from_numpy(out[k * n:(k + 1) * n]).copy_(t[k * n:(k + 1) * n])
            return out
        S_h, Y_h, A_h, d_h, x_h = rows_to_host(wl.S, m), rows_to_host(wl.Y, m), rows_to_host(wl.A, bs), to_np(wl.d), to_np(wl.x)
        g_h, hv_h, tb_h, xs_h, xp_h = np.zeros(n), np.zeros(n), np.zeros(bs), np.zeros(n), x_h.copy()
        rho_r, alpha_r, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
        b_r = _abi.bfgs_mem(S_h.ctypes.data, Y_h.ctypes.data, rho_r.ctypes.data, alpha_r.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, m, m, 3 % m, L, 0.0, 0.0)
        w_r = _abi.workspace_SQN(C.pointer(b_r), dummy.ctypes.data, xs_h.ctypes.data, xp_h.ctypes.data, 0, L, 1, 1, 1, n)
        at_r = {x_h.ctypes.data: x_h, xs_h.ctypes.data: xs_h, xp_h.ctypes.data: xp_h}
        ref = {"be": oracle_backend, "b": b_r, "w": w_r, "req": C.c_void_p(x_h.ctypes.data), "req_vec": C.c_void_p(), "task": C.c_int(101), "info": C.c_int(200),
               "x": x_h.ctypes.data, "g": g_h.ctypes.data, "hv": hv_h.ctypes.data, "log": []}
        dev = {"be": hip_backend, "b": wl.b, "w": wl.w, "req": wl.req, "req_vec": wl.req_vec, "task": wl.task, "info": wl.info,
               "x": wl.x.data_ptr(), "g": wl.grad.data_ptr(), "hv": wl.hv.data_ptr(), "log": []}
        noise_d = torch.empty(n, dtype=torch.float64, device="cuda")

        def gradient(side, t):
            if side is ref:
                np.multiply(d_h, at_r[side["req"].value], out=g_h)
                np.multiply(g_h, noise_h[0], out=g_h)
            else:
                torch.mul(wl.d, wl.ptr2t[side["req"].value], out=wl.grad)
                wl.grad.mul_(noise_d)

        def hess_vec(side):
            if side is ref:
                olib.oracle_fisher_product(A_h.ctypes.data, bs, n, side["req_vec"].value, tb_h.ctypes.data, hv_h.ctypes.data)
            else:
                assert lib.stochqn_hip_fisher_product(wl.A.data_ptr(), bs, n, side["req_vec"].value, wl.t_buf.data_ptr(), wl.hv.data_ptr()) == 0

        noise_h = [None]
        t, new_row = 0, wl.b.mem_st_ix
        while w_r.niter < 2 * L + 1:                                     # niter starts at L: 10 steps, the pair, the step after it
            assert dev["task"].value == ref["task"].value
            if ref["task"].value == 101:
                wl.uniform(noise_d, bench.ST_NOISE, t, 0.99, 0.02)
                noise_h[0] = to_np(noise_d)
                t += 1
            for side in (ref, dev):
                if side["task"].value == 101:
                    gradient(side, t)
                elif side["task"].value == 104:
                    hess_vec(side)
                rc = side["be"].run_SQN(wl.step_size, side["x"], side["g"], side["hv"], C.byref(side["req"]), C.byref(side["req_vec"]),
                                        C.byref(side["task"]), C.byref(side["w"]), C.byref(side["info"]))
                side["log"].append((rc, side["task"].value, side["info"].value, side["w"].niter, side["w"].section, side["b"].mem_used, side["b"].mem_st_ix))
            assert dev["log"][-1] == ref["log"][-1], (len(ref["log"]), dev["log"][-1], ref["log"][-1])
        tasks = [e[1] for e in ref["log"]]
        assert tasks.count(104) == 1 and b_r.mem_st_ix == (new_row + 1) % m and b_r.mem_used == m and all(e[2] == 200 for e in ref["log"])
        assert w_r.niter == 2 * L + 1 and len(ref["log"]) == L + 2       # ten steps, the call that takes the Hessian-vector product in, the step after

        def close(what, got, want, tol=TOL):
            wd = to_dev(want)
            e = float(torch.linalg.vector_norm(got - wd) / torch.linalg.vector_norm(wd))
            print("benchmarked workload, %-12s %.2e from the oracle's" % (what + ":", e))
            assert e <= tol, (what, e)

        close("x", wl.x, x_h)
        close("x_avg_prev", wl.x_avg_prev, xp_h)
        close("new s row", wl.S[new_row * n:(new_row + 1) * n], S_h[new_row * n:(new_row + 1) * n])
        close("new y row", wl.Y[new_row * n:(new_row + 1) * n], Y_h[new_row * n:(new_row + 1) * n])
        assert float(torch.linalg.vector_norm(wl.x_sum - to_dev(xs_h))) <= TOL * float(np.linalg.norm(x_h))       # one step's x after the average was archived
        assert np.allclose(wl.rho_h, rho_r, rtol=1e-9, atol=0) and np.allclose(wl.alpha_h, alpha_r, rtol=1e-7, atol=1e-12 * np.abs(alpha_r).max())
        assert rel_err(x_h, to_np(wl.uniform(noise_d, bench.ST_X0, 0, 1.0, 1.0))) > 1e-3      # and the iterate moved
    finally:
        wl.free()


@pytest.mark.parametrize("config,n", [("c3", 100_000_000), ("c5", 125_000_000)])
def test_bench_headline_workload_runs_clean(config, n):
    """bench.py's own workload under pytest: BASELINE config 3 exactly as measured (SQN n = 1e8, m = 20, L = 10, pairs from
    the 32-row Hessian mini-batch A'(Av)/32, check_nan = 1) and config 5's per-GPU shard (n = 1.25e8): pairs are built and
    accepted, no step is rejected, the objective falls, and the JSON line carries what the driver reads."""
    out = _bench(["--config", config, "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-host-caller", "--sustain-seconds", "1"] + ([] if config == "c3" else ["--no-live-pmc"]), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert ("n=%g per GPU" % n) in d["config"]["workload"] and d["config"]["name"] == config
    assert d["config"]["hess_vec_requests"] == 2 and d["config"]["rejected_steps"] == 0 and d["config"]["rejected_pairs"] == 0
    assert d["config"]["f_end"] < d["config"]["f_start"]
    assert abs(d["value"] - d["steps_per_s_unnormalised"] * n / 1e8) <= 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.5 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel"].split()[0] in ("sdot", "qdot", "sadd")                      # a pass of the default three-pass form dominates
    if config == "c3":                                                             # HBM traffic of that kernel counted in this very run (PMC)
        assert "counted in this run" in r["traffic_source"], r
        assert 0.97 <= r["traffic_over_algorithmic"] <= 1.06 and r["traffic"] == r["traffic_read_bytes"] + r["traffic_write_bytes"]
    assert r["alg_bytes_per_launch"] == (20 + (1 if r["kernel"].startswith("sdot") else 2)) * n * 8
    assert d["two_loop"]["form"] == "three-pass" and d["two_loop"]["bytes_moved"] == (3 * 20 + 5) * n * 8
    assert d["reference_form"]["two_loop_alg_bytes"] == 64 * 20 * n and d["reference_form"]["two_loop_frac_of_8TBps"] > 0.6
    micro = d["two_loop_micro"]
    assert micro["three_pass"]["median_ms"] < micro["sweeps"]["median_ms"]
    assert d["forms"] == dict(d["forms"], three_pass=20, sweeps=0, sweeps_because_of_kappa=0) and d["allreduces_per_step"] == 0
    assert d["sustained"]["steps"] % 10 == 0 and d["sustained"]["seconds"] > 0.5
    assert abs(d["sustained"]["value"] / d["value"] - 1) < 0.15          # the K = 20 steps are representative of a second of the same
    vr = d["value_runs"]                                                 # the K-step region three times, the gradient array re-allocated in between
    assert len(vr["values"]) == 3 and abs(vr["values"][0] - d["value"]) <= 1e-3 * d["value"] + 2e-3 and vr["min"] <= vr["median"] <= vr["max"]
    assert vr["max"] / vr["min"] < 1.25 and d["degraded"] is False and "disjoint supports" in d["config"]["deviation_from_survey_8d"] and d["config"]["workload_resets"] == 0
    if config == "c5":
        assert d["shard_reference_1gpu"]["source"] == "this run"


def test_bench_refuses_to_mislabel_a_smaller_job():
    """--gpus 2 on a box with one GPU and no launcher: exit code != 0, no JSON line (never a 1-GPU run labelled 2)."""
    if torch_cuda().cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=120)
    assert out.returncode != 0
    assert "only 1 device(s) are visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_a_failing_reducer_fails_the_call(hip_backend, capfd):
    """A reduction that fails must not let the step finish on un-reduced local sums (wrong alpha / beta, ranks
    taking different decisions): the call returns -1000 / invalid_input (fault injected through a custom reducer)."""
    import stochqn_amd
    lib = stochqn_amd.cdll()
    REDUCER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
    state = {"fail": False, "calls": 0}

    def reducer(user, buf, count, stream):
        state["calls"] += 1
        return 1 if state["fail"] else 0            # one rank: the sum over ranks is the buffer itself

    keep = REDUCER(reducer)
    lib.stochqn_hip_comm_init_custom.argtypes = [C.c_int, C.c_int, REDUCER, C.c_void_p]
    assert lib.stochqn_hip_comm_init_custom(0, 1, keep, None) == 0
    try:
        P = NoisyQuadratic(800, seed=5)
        opt = OPTIMIZERS["SQN"](backend=hip_backend, space="device", mem_size=4, bfgs_upd_freq=3)
        x = torch_cuda().as_tensor(P.x0(), device="cuda")
        run_trace(opt, P, x, 0.1, 12)
        assert state["calls"] > 0
        state["fail"] = True
        with pytest.raises(ValueError):
            for _ in range(3):                       # at the latest the next step with pairs reduces something
                opt.run_optimizer(x, 0.1)
        assert "instead of continuing on un-reduced sums" in capfd.readouterr().err
    finally:
        lib.stochqn_hip_comm_finalize()


def test_device_errors_fail_the_call_loudly(hip_backend, capfd):
    """A kernel that cannot be launched leaves its outputs stale; handing those back as a result would be
    a silent wrong answer.  Every synchronisation checks the stream and the runtime's last error, and the
    call that saw one returns -1000 with task = invalid_input (fault injected here)."""
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    P = NoisyQuadratic(500, seed=5)
    opt = OPTIMIZERS["SQN"](backend=hip_backend, space="device", mem_size=4, bfgs_upd_freq=3)
    x = torch_cuda().as_tensor(P.x0(), device="cuda")
    run_trace(opt, P, x, 0.1, 10)
    lib.stochqn_hip_set_option(b"inject_device_fault", 1.0)
    with pytest.raises(ValueError):
        opt.run_optimizer(x, 0.1)
    assert "device work failed" in capfd.readouterr().err
    # the raw entry points as well
    from oracle import oracle  # noqa: F401  (only to reuse make_pairs' shapes below)
    rng = np.random.default_rng(0)
    S, Y = make_pairs(rng, 300, 3)
    dS, dY, dg = (torch_cuda().as_tensor(a, device="cuda") for a in (S, Y, rng.random(300)))
    lib.stochqn_hip_set_option(b"inject_device_fault", 1.0)
    rho, alpha = np.zeros(3), np.zeros(3)
    lib.stochqn_hip_two_loop.restype = C.c_int
    lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    assert lib.stochqn_hip_two_loop(dg.data_ptr(), 300, None, 0.0, dY.data_ptr(), dS.data_ptr(), 3, 3, 0,
                                    rho.ctypes.data, alpha.ctypes.data) == -1000
    assert lib.stochqn_hip_two_loop(dg.data_ptr(), 300, None, 0.0, dY.data_ptr(), dS.data_ptr(), 3, 3, 0,
                                    rho.ctypes.data, alpha.ctypes.data) == 0
    lib.stochqn_hip_release(C.c_void_p(dS.data_ptr()))


def synth_u(i, seed, stream, t):
    """numpy restatement of the counter-based generator of include/stochqn_hip.h (uint64 arithmetic mod 2^64)."""
    M = np.uint64
    with np.errstate(over="ignore"):
        key = M(seed) ^ (M(stream) * M(0x9E3779B97F4A7C15)) ^ (M(t) * M(0xD1B54A32D192ED03))
        z = key + i.astype(np.uint64) * M(0x9E3779B97F4A7C15)
        z = (z ^ (z >> M(30))) * M(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> M(27))) * M(0x94D049BB133111EB)
        z = z ^ (z >> M(31))
    return (z >> M(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@pytest.mark.parametrize("count,first", [(1, 0), (7, 5), (1000, 123456789012), (100003, 3)])
def test_counter_based_inputs_match_their_definition(count, first, hip_backend):
    """The synthetic-input kernels against the formula in the header, bit for bit; and a slice generated on its own
    equals the same slice of a longer vector (what makes the inputs shard-invariant)."""
    import stochqn_amd
    torch = torch_cuda()
    lib = stochqn_amd.cdll()
    u64 = C.c_ulonglong
    lib.stochqn_hip_synth_uniform.argtypes = [C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double, C.c_double]
    lib.stochqn_hip_synth_noisy_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double]
    lib.stochqn_hip_synth_batch_row.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, u64, C.c_uint, C.c_uint]
    seed = 20240611
    idx = np.arange(first, first + count, dtype=np.uint64)
    out = torch.empty(count, dtype=torch.float64, device="cuda")
    assert lib.stochqn_hip_synth_uniform(out.data_ptr(), count, first, seed, 3, 7, 0.5, 2.0) == 0
    assert np.array_equal(out.cpu().numpy(), 0.5 + 2.0 * synth_u(idx, seed, 3, 7))
    d = torch.as_tensor(0.5 + synth_u(idx, seed, 0, 0), device="cuda")
    x = torch.as_tensor(1.0 + synth_u(idx, seed, 3, 0), device="cuda")
    g = torch.empty(count, dtype=torch.float64, device="cuda")
    assert lib.stochqn_hip_synth_noisy_grad(g.data_ptr(), d.data_ptr(), x.data_ptr(), count, first, seed, 4, 11, 0.01) == 0
    want = (d.cpu().numpy() * x.cpu().numpy()) * (1.0 + 0.01 * (2.0 * synth_u(idx, seed, 4, 11) - 1.0))
    assert np.array_equal(g.cpu().numpy(), want)
    row = torch.empty(count, dtype=torch.float64, device="cuda")
    assert lib.stochqn_hip_synth_batch_row(row.data_ptr(), d.data_ptr(), count, first, 3, 32) == 0
    want = np.where((idx % np.uint64(32)) == 3, np.sqrt(32.0 * d.cpu().numpy()), 0.0)
    assert np.array_equal(row.cpu().numpy(), want)
    if count > 10:                                              # a slice on its own = the slice of the whole
        part = torch.empty(count - 5, dtype=torch.float64, device="cuda")
        assert lib.stochqn_hip_synth_uniform(part.data_ptr(), count - 5, first + 5, seed, 3, 7, 0.5, 2.0) == 0
        assert torch.equal(part, out[5:])
