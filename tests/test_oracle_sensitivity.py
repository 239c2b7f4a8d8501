"""Where the GPU parity tests loosen north_star's 1e-10 for a FREE-RUNNING trajectory (tests/test_gpu_parity.py: FREE_RUN_TOL), the
looseness must be the algorithm's own -- shown on the CPU, with the kernels out of the question: the oracle
against ITSELF with nothing changed but the order in which its dot products are summed (every one an order a BLAS is free to
choose for the reference's cblas_ddot / dgemv, src/stochqn.c:676-706, 936-952).  tools/oracle_sensitivity.py wrote
profiles/r06_oracle_vs_oracle_sensitivity.json, tools/free_run_report.py (on an MI355X) profiles/r06_free_run_device_vs_oracle.json."""
import json
import os
import re

import numpy as np
import pytest

from oracle import oracle
from harness import NoisyQuadratic, OPTIMIZERS, INT_KEYS, rel_err, run_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-10


def free_run_tol():
    """FREE_RUN_TOL of tests/test_gpu_parity.py, read from its text (the module itself needs a GPU to be of any use)."""
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    m = re.search(r"^FREE_RUN_TOL = (\{[^}]*\})", src, re.M)
    return eval(m.group(1))


def _distance(optname, kw, n, step, calls, lanes):
    out = []
    for ln in (8, lanes):
        oracle.cdll().oracle_set_lanes(ln)
        try:
            P = NoisyQuadratic(n, seed=7)
            out.append(run_trace(OPTIMIZERS[optname](backend=oracle.bound(), space="host", **kw), P, P.x0(), step, calls))
        finally:
            oracle.cdll().oracle_set_lanes(8)
    base, other = out
    assert all(o[k] == b[k] for o, b in zip(other, base) for k in INT_KEYS if k in b)          # every discrete output identical
    return max(max(rel_err(o["x"], b["x"]), rel_err(o["req"], b["req"])) for o, b in zip(other, base))


def test_two_summation_orders_of_the_oracle_drift_apart_exactly_where_the_gpu_bar_is_loose():
    """Live, at the sizes where it shows: adaQN with AdaGrad scaling and Fisher pairs (FREE_RUN_TOL 1e-7) -- two orders of the same
    dot products end 1e-9 .. 1e-7 apart at n = 70001 --, adaqn_ring20 (2e-9) stays under its bar, and the controls (a 20-pair SQN
    ring, adaqn_ring25: both held to 1e-10 on the GPU) stay orders of magnitude below 1e-10."""
    from tools.oracle_sensitivity import CONFIGS
    tol = free_run_tol()
    loose = _distance(*CONFIGS["adaqn_fisher_adagrad_nomaxincr"][:2], 70001, *CONFIGS["adaqn_fisher_adagrad_nomaxincr"][2:], lanes=1)
    assert 1e-9 < loose <= 3 * tol["adaqn_fisher_adagrad_nomaxincr"], loose
    r20 = _distance(*CONFIGS["adaqn_ring20"][:2], 4097, *CONFIGS["adaqn_ring20"][2:], lanes=4)
    assert r20 <= tol["adaqn_ring20"], r20
    assert "adaqn_ring25" not in tol and "sqn_ring20" not in tol
    assert _distance(*CONFIGS["adaqn_ring25"][:2], 4097, *CONFIGS["adaqn_ring25"][2:], lanes=1) <= 1e-12
    assert _distance(*CONFIGS["sqn_ring20"][:2], 4097, *CONFIGS["sqn_ring20"][2:], lanes=1) <= 1e-13


def test_the_committed_sensitivity_tables_back_every_loosened_bar_and_only_those():
    tol = free_run_tol()
    cpu = json.load(open(os.path.join(ROOT, "profiles", "r06_oracle_vs_oracle_sensitivity.json")))["worst_free_running_rel_err_by_config"]
    gpu = json.load(open(os.path.join(ROOT, "profiles", "r06_free_run_device_vs_oracle.json")))["worst_free_running_rel_err_by_config"]
    assert set(cpu) == set(gpu) and set(tol) <= set(cpu)
    for name in cpu:
        if name in tol:
            # loosened only where the oracle disagrees with itself beyond 1e-10, and the device is no further from the oracle than
            # a few times what the oracle's own summation orders are from each other
            assert cpu[name] > TOL, (name, cpu[name])
            assert gpu[name] <= tol[name] and gpu[name] <= 4 * cpu[name], (name, gpu[name], cpu[name])
        else:
            assert cpu[name] <= TOL and gpu[name] <= TOL, (name, cpu[name], gpu[name])
    # every run in both tables took the same decisions as its baseline
    for f in ("r06_oracle_vs_oracle_sensitivity.json", "r06_free_run_device_vs_oracle.json"):
        assert all(r["discrete_outputs_identical"] for r in json.load(open(os.path.join(ROOT, "profiles", f)))["runs"])
