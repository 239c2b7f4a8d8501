#!/usr/bin/env python3
"""How far do two LEGAL evaluations of the reference's arithmetic drift apart along a free-running adaQN trajectory?

VERDICT r05 #3: the GPU parity tests hold three adaQN configurations (and the n = 1e8 free-running instance) to 1e-7 instead of
north_star's 1e-10 -- "explained by amplification / cancellation, but nowhere SHOWN to be the algorithm's own sensitivity rather
than the kernels'".  This tool takes the kernels out of the question: the CPU oracle against ITSELF, with nothing changed but the
order in which its dot products are summed -- every one of them an order a BLAS is free to choose (reference src/stochqn.c calls
cblas_ddot / dgemv of whatever library it was linked with; SURVEY.md 8c: "only summation order is implementation-defined"):

    lanes8   the oracle's default: 8 interleaved partial sums (what the GPU tests compare with)
    lanes4   4 interleaved partial sums
    lanes1   the textbook sequential sum
    openblas the OpenBLAS that scipy / numpy bundle (oracle_use_cblas), where one is found

For each configuration and problem size of tests/test_gpu_parity.py it drives the same noisy quadratic through the free-mode
object over the oracle and records, per call, the relative distance of x (and of the requested vector) from the lanes8 run, and
whether every discrete output (task, info, counters) is identical.  Also: the C4-shaped instance of
test_adaqn_trajectory_at_full_size_in_lockstep_with_the_oracle (m = 20, fisher 16, L = 1, step 0.002, RMSProp) at reduced n.

    python tools/oracle_sensitivity.py > profiles/r06_oracle_vs_oracle_sensitivity.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from oracle import oracle
from harness import NoisyQuadratic, OPTIMIZERS, rel_err, run_trace, INT_KEYS

CONFIGS = {
    "adaqn_fisher_adagrad_nomaxincr": ("adaQN", dict(mem_size=3, fisher_size=5, bfgs_upd_freq=3, max_incr=None), 0.05, 70),
    "adaqn_ring25": ("adaQN", dict(mem_size=25, fisher_size=8, bfgs_upd_freq=1, max_incr=None, min_curvature=None), 0.002, 31),
    "adaqn_ring20": ("adaQN", dict(mem_size=20, fisher_size=16, bfgs_upd_freq=1, max_incr=None, min_curvature=None, rmsprop_weight=0.9), 0.002, 30),
    # controls: configurations the GPU tests hold to 1e-10 free-running
    "sqn_ring20": ("SQN", dict(mem_size=20, bfgs_upd_freq=1, min_curvature=None), 0.05, 60),
    "adaqn_fisher_rms": ("adaQN", dict(mem_size=3, fisher_size=7, bfgs_upd_freq=4, rmsprop_weight=0.9), 0.05, 90),
    # the C4-shaped instance of the n = 1e8 lock-step test, at reduced n (22 iterations there; L = 1: one call per iteration + 1)
    "c4_lockstep_instance": ("adaQN", dict(mem_size=20, fisher_size=16, bfgs_upd_freq=1, max_incr=None, min_curvature=None, rmsprop_weight=0.9), 0.002, 27),
}
SIZES = {"default": (7, 64, 1000, 4097, 70001), "c4_lockstep_instance": (100_000, 1_000_000)}


def variants():
    yield "lanes4", lambda: oracle.cdll().oracle_set_lanes(4), lambda: oracle.cdll().oracle_set_lanes(8)
    yield "lanes1", lambda: oracle.cdll().oracle_set_lanes(1), lambda: oracle.cdll().oracle_set_lanes(8)
    info = oracle.find_openblas()
    if info is not None:
        yield "openblas", lambda: oracle.use_cblas(info, threads=2), lambda: oracle.use_cblas(None)


def trace(optname, kw, n, step, calls, seed):
    P = NoisyQuadratic(n, seed=seed)
    return run_trace(OPTIMIZERS[optname](backend=oracle.bound(), space="host", **kw), P, P.x0(), step, calls)


def sensitivity(name, seed=7):
    optname, kw, step, calls = CONFIGS[name]
    out = []
    for n in SIZES.get(name, SIZES["default"]):
        oracle.cdll().oracle_set_lanes(8)
        oracle.use_cblas(None)
        base = trace(optname, kw, n, step, calls, seed)
        for vname, on, off in variants():
            on()
            try:
                tr = trace(optname, kw, n, step, calls, seed)
            finally:
                off()
            per_call = [max(rel_err(t["x"], b["x"]), rel_err(t["req"], b["req"])) for t, b in zip(tr, base)]
            same = all(t[k] == b[k] for t, b in zip(tr, base) for k in INT_KEYS if k in b)
            first = next((i for i, e in enumerate(per_call) if e > 1e-10), None)
            out.append({"config": name, "n": n, "variant": vname, "calls": calls, "discrete_outputs_identical": bool(same),
                        "max_rel_err": max(per_call), "final_rel_err": per_call[-1], "first_call_above_1e-10": first,
                        "per_call": [float("%.3e" % e) for e in per_call]})
    return out


def full_size(n=100_000_000, calls=27):
    """The C4-shaped instance at n = 1e8 itself: 26 iterations (the length of test_full_size_steps_agree_between_the_two_forms'
    adaQN case), the oracle's default order against 4 interleaved partial sums.  ~60 GB of host memory, minutes on 16 cores:
    run where there is room (the GPU box), not in the test suite."""
    optname, kw, step, _ = CONFIGS["c4_lockstep_instance"]
    oracle.set_threads(oracle.usable_cpus())
    rng = np.random.default_rng(99)
    d = 0.5 + rng.random(n)
    dn = [d * (1 + 0.01 * (2 * rng.random(n) - 1)) for _ in range(2)]
    x0 = 1 + rng.random(n)
    xs = {}
    for lanes in (8, 4):
        oracle.cdll().oracle_set_lanes(lanes)
        opt = OPTIMIZERS[optname](backend=oracle.bound(), space="host", **kw)
        x, t = x0.copy(), 0
        for _ in range(calls):
            r = opt.run_optimizer(x, step)
            assert r["task"] == "calc_grad"
            np.multiply(dn[t % 2], r["requested_on"], out=opt.gradient)
            t += 1
        xs[lanes] = (x, opt.niter, opt.BFGS_mem.mem_used)
        sys.stderr.write("n = %g, lanes %d: %d iterations done\n" % (n, lanes, opt.niter))
    oracle.cdll().oracle_set_lanes(8)
    (xa, ia, ma), (xb, ib, mb) = xs[8], xs[4]
    return {"config": "c4_lockstep_instance", "n": n, "variant": "lanes4", "calls": calls, "iterations": ia,
            "discrete_outputs_identical": bool(ia == ib and ma == mb), "final_rel_err": rel_err(xb, xa),
            "moved": rel_err(xa, x0)}


if __name__ == "__main__":
    if sys.argv[1:2] == ["full_size"]:
        json.dump(full_size(int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000), sys.stdout)
        sys.stdout.write("\n")
        sys.exit(0)
    which = sys.argv[1:] or list(CONFIGS)
    oracle.set_threads(2)
    res = []
    for name in which:
        res += sensitivity(name)
        sys.stderr.write("%s done\n" % name)
    worst = {}
    for r in res:
        worst[r["config"]] = max(worst.get(r["config"], 0.0), r["max_rel_err"])
    json.dump({"what": __doc__.split("\n\n")[0] + "  The CPU oracle against itself, summation order of its dot products changed, nothing else.",
               "worst_free_running_rel_err_by_config": worst, "runs": res}, sys.stdout, indent=1)
    sys.stdout.write("\n")
