"""Generates tests/golden/traces.json: per-call traces of the PINNED CPU ORACLE on scripted noisy
quadratics (every section of every optimiser, ring wrap, NaN gradient, rejected pair,
func_increased, hess_init, y_reg, AdaGrad vs RMSProp, grad-diff vs Hess-vec vs Fisher).

These are regression vectors for both the oracle and the HIP library.  They are NOT outputs of the
reference: the reference cannot be built under the project rules (DESIGN.md "oracle"); the oracle
that produced them is pinned to the reference by tests/golden/known_answers.json.

    python tests/golden/make_traces.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from harness import OPTIMIZERS, NoisyQuadratic, run_trace  # noqa: E402
from oracle import oracle  # noqa: E402
from test_gpu_parity import CONFIGS  # noqa: E402

N = 7


def main():
    be = oracle.bound()
    out = {"_about": __doc__, "traces": {}}
    for name, optname, kw, step, calls, pkw in CONFIGS:
        P = NoisyQuadratic(N, seed=7, **pkw)
        opt = OPTIMIZERS[optname](backend=be, space="host", **kw)
        tr = run_trace(opt, P, P.x0(), step, calls)
        for r in tr:
            for k, v in list(r.items()):
                if hasattr(v, "tolist"):
                    r[k] = [None if x != x else x for x in v.tolist()]
        out["traces"][name] = {"n": N, "trace": tr}
    with open(os.path.join(HERE, "traces.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(out["traces"]), "traces")


if __name__ == "__main__":
    main()
