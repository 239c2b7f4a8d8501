#!/usr/bin/env python3
"""tools/oracle_long_run.py -- bench.py's headline workload (SQN, m = 20, L = 10, Hv = A'(Av)/32, ring pre-filled and wrapped,
check_nan = 1) through the CPU ORACLE (oracle_run_SQN + oracle_fisher_product; reference src/stochqn.c:1051-1115, 936-952) for
thousands of steps with ONE optimiser state, at an n the CPU gets through: does the oracle stall and diverge where the device
library does (profiles/r04b_long_run_ladder.log, VERDICT r04 weak #8 / task 6)?  No GPU involved: the synthetic inputs are the
numpy restatement of the counter-based generator of include/stochqn_hip.h (tests/test_gpu_parity.py checks the kernels
against the same formula bit for bit), so the device run of the same instance -- `bench.py --vars-per-gpu N --steps K` -- starts
from the same bits.

    python tools/oracle_long_run.py --n 1000000 --steps 3000 --every 250 --batch rank32
    python tools/oracle_long_run.py --n 1000000 --steps 6000 --every 500 --batch averaged

--batch rank32    the batch of rounds 1 - 4: a_k,i = sqrt(bs d_i) on i = k mod bs.  A'A/bs has bs eigenvalues of about n/bs: the
                  curvature pairs say "steep" by a factor n/bs, gamma = s'y/y'y is ~ bs/n and the step collapses.
--batch averaged  round 5: the same supports, a_k,i = bs sqrt(d_i / n): A'A/bs = D^(1/2) P D^(1/2) with P the projector onto the
                  bs class indicators -- eigenvalues ~ mean(d), below the true Hessian diag(d) in the Loewner order.
Prints one line per rung: step, f = 1/2 sum d x^2, |x|, rejected steps / pairs so far, ring use.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20240611
ST_D, ST_S, ST_X0, ST_NOISE = 0, 1, 3, 4


def synth_u(i, stream, t):
    M = np.uint64
    with np.errstate(over="ignore"):
        key = M(SEED) ^ (M(stream) * M(0x9E3779B97F4A7C15)) ^ (M(t) * M(0xD1B54A32D192ED03))
        z = key + i * M(0x9E3779B97F4A7C15)
        z = (z ^ (z >> M(30))) * M(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> M(27))) * M(0x94D049BB133111EB)
        z = z ^ (z >> M(31))
    return (z >> M(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def batch_rows(d, bs, kind, n_total):
    """The Hessian mini-batch exactly as bench.py's Workload makes it (stochqn_hip_synth_batch_row on d, or on d * bs / n)."""
    n = d.size
    idx = np.arange(n, dtype=np.uint64)
    base = d if kind == "rank32" else d * (bs / float(n_total))
    A = np.zeros(bs * n)
    for k in range(bs):
        A[k * n:(k + 1) * n] = np.where((idx % np.uint64(bs)) == k, np.sqrt(float(bs) * base), 0.0)
    return A


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=20)
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--step-size", type=float, default=0.05)
    ap.add_argument("--batch", default="rank32", choices=["rank32", "averaged"])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    from oracle import oracle
    from stochqn_amd import _abi
    be, olib = oracle.bound(), oracle.cdll()
    oracle.set_threads(a.threads if a.threads > 0 else oracle.usable_cpus())
    n, m, L, bs = a.n, a.m, a.L, a.bs
    idx = np.arange(n, dtype=np.uint64)
    d = 0.5 + 1.0 * synth_u(idx, ST_D, 0)
    x = 1.0 + 1.0 * synth_u(idx, ST_X0, 0)
    A = batch_rows(d, bs, a.batch, n)
    S, Y = np.empty(m * n), np.empty(m * n)
    for k in range(m):
        S[k * n:(k + 1) * n] = -0.5e-3 + 1e-3 * synth_u(idx, ST_S, k)
        Y[k * n:(k + 1) * n] = d * S[k * n:(k + 1) * n]
    grad, hv, tb = np.zeros(n), np.zeros(n), np.zeros(bs)
    x_sum, x_avg_prev = np.zeros(n), x.copy()
    rho_h, alpha_h, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho_h.ctypes.data, alpha_h.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, m, m, 3 % m, L, 0.0, 0.0)
    w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, L, 1, 1, 1, n)
    req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)
    at = {x.ctypes.data: x, x_sum.ctypes.data: x_sum, x_avg_prev.ctypes.data: x_avg_prev}
    bad = rejected = 0
    t0 = time.time()
    rungs = []

    def rung(step):
        f = 0.5 * float(np.sum(d * x * x))
        r = {"step": step, "f": f, "norm_x": float(np.linalg.norm(x)), "rejected_steps": bad, "rejected_pairs": rejected, "mem_used": int(b.mem_used), "seconds": round(time.time() - t0, 1)}
        rungs.append(r)
        print("%6d  f %.17g  |x| %.6g  rejected steps %d pairs %d  ring %d  (%.0f s)" % (step, f, r["norm_x"], bad, rejected, b.mem_used, r["seconds"]), flush=True)
        return f

    f0 = rung(0)
    t = 0
    for step in range(1, a.steps + 1):
        target = w.niter + 1
        while w.niter < target:
            if task.value == 101:
                noise = 1.0 + 0.01 * (2.0 * synth_u(idx, ST_NOISE, t) - 1.0)
                np.multiply(d, at[req.value], out=grad)
                np.multiply(grad, noise, out=grad)
            elif task.value == 104:
                olib.oracle_fisher_product(A.ctypes.data, bs, n, req_vec.value, tb.ctypes.data, hv.ctypes.data)
            rc = be.run_SQN(a.step_size, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
            assert rc in (0, 1), rc
            bad += info.value == 203
            rejected += info.value == 202
        t += 1
        if step % a.every == 0 or step == a.steps:
            f = rung(step)
            if not np.isfinite(f) or f > 1e6 * f0:
                print("diverged")
                break
    if a.json:
        with open(a.json, "w") as fh:
            json.dump({"workload": "SQN n=%d m=%d L=%d bs=%d step %.3g, batch '%s', ring pre-filled (y = d s), mem_st_ix = 3, check_nan = 1" % (n, m, L, bs, a.step_size, a.batch),
                       "backend": "oracle/stochqn_oracle.c (oracle_run_SQN + oracle_fisher_product)", "rungs": rungs}, fh, indent=1)


if __name__ == "__main__":
    main()
