"""The oracle against the committed regression traces, and oracle-internal consistency checks
(threads do not change results beyond rounding; library-owned vs caller-owned workspaces agree)."""
import json
import os

import numpy as np
import pytest

from harness import OPTIMIZERS, NoisyQuadratic, compare_traces, rel_err, run_trace
from test_gpu_parity import CONFIGS, make_pairs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "traces.json")


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_oracle_matches_golden_traces(cfg, oracle_backend):
    gold = json.load(open(GOLD))["traces"][cfg[0]]
    P = NoisyQuadratic(gold["n"], seed=7, **cfg[5])
    opt = OPTIMIZERS[cfg[1]](backend=oracle_backend, space="host", **cfg[2])
    got = run_trace(opt, P, P.x0(), cfg[3], cfg[4])
    compare_traces(got, gold["trace"], 1e-13)


def test_golden_traces_cover_every_branch():
    gold = json.load(open(GOLD))["traces"]
    infos, tasks, sections = set(), set(), set()
    for e in gold.values():
        for r in e["trace"]:
            infos.add(r["info"]); tasks.add(r["task"]); sections.add(r["section"])
    assert infos == {"no_problems_encountered", "func_increased", "curvature_too_small", "search_direction_was_nan"}
    assert tasks == {"calc_grad", "calc_grad_same_batch", "calc_grad_big_batch", "calc_hess_vec", "calc_fun_val_batch"}
    assert sections == {1, 2, 3, 4, 5}


def test_two_loop_threads_agree():
    from oracle import oracle
    n, m = 300_000, 6
    rng = np.random.default_rng(0)
    S, Y = make_pairs(rng, n, m)
    g = rng.random(n) - 0.5
    a, b = g.copy(), g.copy()
    oracle.set_threads(1)
    oracle.two_loop(a, None, 0.0, Y, S, m, m, 2)
    oracle.set_threads(4)
    oracle.two_loop(b, None, 0.0, Y, S, m, m, 2)
    oracle.set_threads(1)
    assert rel_err(b, a) < 1e-12


def test_two_loop_satisfies_secant_equation():
    """H_k y_{k-1} = s_{k-1}: property used by the full-size GPU tests, checked on the oracle."""
    from oracle import oracle
    n, m, st = 5000, 7, 4
    rng = np.random.default_rng(1)
    S, Y = make_pairs(rng, n, m)
    newest = (st + m - 1) % m
    q = Y[newest * n:(newest + 1) * n].copy()
    oracle.two_loop(q, None, 0.0, Y, S, m, m, st)
    assert rel_err(q, S[newest * n:(newest + 1) * n]) < 1e-11


def test_float_oracle_tracks_double_oracle():
    """The -DUSE_FLOAT build of the oracle (checker of libstochqn_f32.so) follows the double one to
    float accuracy on a contractive problem, with identical request sequences."""
    from oracle import oracle
    cfg = [c for c in CONFIGS if c[0] == "sqn_hessvec"][0]
    P = NoisyQuadratic(200, seed=7)
    d = run_trace(OPTIMIZERS["SQN"](backend=oracle.bound(), **cfg[2]), P, P.x0(), cfg[3], 40)
    f = run_trace(OPTIMIZERS["SQN"](backend=oracle.bound_f32(), use_float=True, **cfg[2]), P, P.x0().astype(np.float32), cfg[3], 40)
    assert [r["task"] for r in f] == [r["task"] for r in d]
    assert rel_err(f[-1]["x"], d[-1]["x"]) < 1e-4
