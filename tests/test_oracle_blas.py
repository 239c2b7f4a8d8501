"""The oracle with its BLAS-1/2 calls routed through a real CBLAS (the OpenBLAS that scipy / numpy bundle) -- what bench.py's
`cpu_baseline.blas` times ("src/stochqn.c + BLAS", reference src/stochqn.c:676-706, 946-949) -- computes what
the oracle's own loops compute, up to the summation order of the library.  Skipped where no such library is found."""
import numpy as np
import pytest

from oracle import oracle
from harness import NoisyQuadratic, compare_traces, run_trace
import stochqn_amd


@pytest.fixture()
def blas():
    info = oracle.find_openblas()
    if info is None:
        pytest.skip("no OpenBLAS with a CBLAS interface on this box")
    got = oracle.use_cblas(info, threads=2)
    assert got is not None and got["library"] == info["path"]
    yield got
    oracle.use_cblas(None)
    assert oracle.cdll().oracle_uses_cblas() == 0


def _two_loop_inputs(n, m, seed=5):
    rng = np.random.default_rng(seed)
    d = 0.5 + rng.random(n)
    S = 1e-3 * (rng.random((m, n)) - 0.5)
    Y = (S * d).reshape(-1)
    return S.reshape(-1), Y, rng.random(n) - 0.5


@pytest.mark.parametrize("which", ["lp64", "ilp64"])
def test_both_bundled_libraries_give_the_oracles_two_loop_and_fisher_product(which):
    import glob
    import os
    import site
    want = {"lp64": ("scipy.libs", "scipy_%s", 0), "ilp64": ("numpy.libs", "scipy_%s64_", 1)}[which]
    found = [f for root in site.getsitepackages() for f in glob.glob(os.path.join(root, want[0], "libscipy_openblas*.so"))]
    if not found:
        pytest.skip("%s is not bundled here" % want[0])
    n, m, st = 300_007, 7, 3
    S, Y, g = _two_loop_inputs(n, m)
    own = g.copy()
    rho0, al0 = oracle.two_loop(own, None, 0.0, Y, S, m, m, st)
    F = np.random.default_rng(9).random((11, n)) - 0.5
    s = g.copy()
    t0, y0 = oracle.fisher_product(F.reshape(-1), 11, s)
    try:
        assert oracle.use_cblas({"path": found[0], "fmt": want[1], "ilp64": want[2]}, threads=2) is not None
        assert oracle.cdll().oracle_uses_cblas() == 1
        via = g.copy()
        rho1, al1 = oracle.two_loop(via, None, 0.0, Y, S, m, m, st)
        t1, y1 = oracle.fisher_product(F.reshape(-1), 11, s)
    finally:
        oracle.use_cblas(None)
    assert np.linalg.norm(via - own) <= 1e-12 * np.linalg.norm(own)
    assert np.allclose(rho1, rho0, rtol=1e-12) and np.allclose(al1, al0, rtol=1e-10, atol=1e-18)
    assert np.linalg.norm(t1 - t0) <= 1e-12 * np.linalg.norm(t0) and np.linalg.norm(y1 - y0) <= 1e-12 * np.linalg.norm(y0)


@pytest.mark.parametrize("opt,kw", [("SQN_free", dict(mem_size=4, bfgs_upd_freq=3)),
                                    ("oLBFGS_free", dict(mem_size=5)),
                                    ("adaQN_free", dict(mem_size=4, fisher_size=6, bfgs_upd_freq=3, rmsprop_weight=0.9, use_grad_diff=False))])
def test_whole_trajectories_through_the_blas_match_the_own_loops(blas, opt, kw):
    P = NoisyQuadratic(4000, seed=11)
    oracle.use_cblas(None)
    want = run_trace(getattr(stochqn_amd, opt)(backend=oracle.bound(), **kw), P, P.x0(), 0.05, 45)
    assert oracle.use_cblas(oracle.find_openblas(), threads=2) is not None
    got = run_trace(getattr(stochqn_amd, opt)(backend=oracle.bound(), **kw), P, P.x0(), 0.05, 45)
    compare_traces(got, want, 1e-10)
