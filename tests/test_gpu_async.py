"""Option "async_device": stream-ordered run_* for device-resident callers of configurations in which no decision depends on
device data (check_nan = 0, min_curvature = 0).  The kernels are the same, only the waiting is gone: every x, every request
and every counter must equal the synchronous run bit for bit; configurations that can reject something keep synchronising."""
import ctypes as C

import numpy as np
import pytest

from harness import NoisyQuadratic, OPTIMIZERS, run_trace

pytestmark = pytest.mark.gpu

KW = {
    "oLBFGS": dict(mem_size=5, min_curvature=None, check_nan=False),
    "SQN": dict(mem_size=5, bfgs_upd_freq=3, min_curvature=None, check_nan=False),
    "SQN_graddiff": dict(mem_size=4, bfgs_upd_freq=3, min_curvature=None, check_nan=False, use_grad_diff=True),
    "adaQN": dict(mem_size=5, fisher_size=7, bfgs_upd_freq=3, max_incr=1.01, min_curvature=None, rmsprop_weight=0.9, check_nan=False),
}


def _lib():
    import stochqn_amd
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    return lib


@pytest.mark.parametrize("name", sorted(KW))
@pytest.mark.parametrize("n", [500, 300_001])
def test_stream_ordered_calls_equal_synchronous_calls(name, n, hip_backend):
    import torch
    lib = _lib()
    P = NoisyQuadratic(n, seed=13)
    out = []
    try:
        for flag in (0.0, 1.0):
            assert lib.stochqn_hip_set_option(b"async_device", flag) == 0
            x = torch.as_tensor(P.x0(), device="cuda:0")
            opt = OPTIMIZERS[name.split("_")[0]](backend=hip_backend, space="device", device="cuda:0", **KW[name])
            out.append(run_trace(opt, P, x, 0.05, 40))
            lib.stochqn_hip_release_all()
    finally:
        lib.stochqn_hip_set_option(b"async_device", 0.0)
    for i, (a, b) in enumerate(zip(*out)):
        for k in ("task", "info", "changed", "niter", "section", "mem_used", "mem_st_ix", "req_id"):
            assert a[k] == b[k], (i, k)
        for k in ("x", "req", "req_vec"):
            if k in a:
                assert np.array_equal(a[k], b[k]), (i, k)


def test_guarded_configurations_keep_synchronising(hip_backend, oracle_backend):
    """With the option on, a configuration that CAN reject a step (check_nan = 1) or a pair (min_curvature > 0) still takes
    the synchronous path: NaN gradients and flat pairs get the oracle's verdicts."""
    import torch
    from harness import compare_traces
    lib = _lib()
    P = NoisyQuadratic(2000, seed=4, nan_calls=(9,))
    kw = dict(mem_size=4, bfgs_upd_freq=3, min_curvature=1e-4)
    want = run_trace(OPTIMIZERS["SQN"](backend=oracle_backend, space="host", **kw), P, P.x0(), 0.05, 30)
    try:
        assert lib.stochqn_hip_set_option(b"async_device", 1.0) == 0
        x = torch.as_tensor(P.x0(), device="cuda:0")
        got = run_trace(OPTIMIZERS["SQN"](backend=hip_backend, space="device", device="cuda:0", **kw), P, x, 0.05, 30)
    finally:
        lib.stochqn_hip_set_option(b"async_device", 0.0)
        lib.stochqn_hip_release_all()
    compare_traces(got, want, 1e-10)
    assert any(r["info"] == "search_direction_was_nan" for r in got)
