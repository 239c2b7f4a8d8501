#!/usr/bin/env python3
"""GPU library against the CPU oracle, FREE-RUNNING, on the configurations whose bar the parity tests loosen to 1e-7
(tests/test_gpu_parity.py: FREE_RUN_TOL) and two controls: the worst relative error of x / the requested vector along the
trajectory, per configuration, problem size and form of the two-loop recursion.  Next to profiles/r06_oracle_vs_oracle_
sensitivity.json (the oracle against itself, summation order changed) this says whether the distance between device and oracle
is the distance between any two legal evaluations of the reference's arithmetic.  Run on the GPU box:

    python tools/free_run_report.py > gpurun_out/r06/free_run_report.json
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch

import stochqn_amd
from oracle import oracle
from harness import NoisyQuadratic, OPTIMIZERS, INT_KEYS, rel_err, run_trace
from tools.oracle_sensitivity import CONFIGS, SIZES

lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
hip = stochqn_amd.lib()
out = []
for name, (optname, kw, step, calls) in CONFIGS.items():
    for n in SIZES.get(name, SIZES["default"]):
        P = NoisyQuadratic(n, seed=7)
        want = run_trace(OPTIMIZERS[optname](backend=oracle.bound(), space="host", **kw), P, P.x0(), step, calls)
        for form, three in (("three_pass", 1.0), ("sweeps", 0.0)):
            assert lib.stochqn_hip_set_option(b"threepass", three) == 0
            opt = OPTIMIZERS[optname](backend=hip, space="device", **kw)
            got = run_trace(opt, P, torch.as_tensor(P.x0(), device="cuda"), step, calls)
            opt.release()
            per = [max(rel_err(g["x"], w["x"]), rel_err(g["req"], w["req"])) for g, w in zip(got, want)]
            same = all(g[k] == w[k] for g, w in zip(got, want) for k in INT_KEYS if k in w)
            out.append({"config": name, "n": n, "form": form, "calls": calls, "discrete_outputs_identical": bool(same), "max_rel_err": max(per),
                        "first_call_above_1e-10": next((i for i, e in enumerate(per) if e > 1e-10), None), "per_call": [float("%.3e" % e) for e in per]})
            sys.stderr.write("%s n=%d %s: %.2e\n" % (name, n, form, max(per)))
lib.stochqn_hip_set_option(b"threepass", 1.0)
worst = {}
for r in out:
    worst[r["config"]] = max(worst.get(r["config"], 0.0), r["max_rel_err"])
json.dump({"what": "libstochqn.so on an MI355X against the CPU oracle, free-running, per call max of rel err of x and of the requested vector",
           "worst_free_running_rel_err_by_config": worst, "runs": out}, sys.stdout, indent=1)
sys.stdout.write("\n")
