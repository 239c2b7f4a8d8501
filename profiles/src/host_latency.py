"""Per-step wall time of a HOST caller (numpy arrays, profile B) versus n; scratch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochqn_amd
from stochqn_amd import SQN_free
for n in (1000, 10000, 100000, 1000000, 10000000):
    rng = np.random.default_rng(1)
    d = 0.5 + rng.random(n); x = 1.0 + rng.random(n)
    opt = SQN_free(mem_size=10, bfgs_upd_freq=10, min_curvature=None, space="host")
    def advance(k):
        target = (opt.niter if opt.initialized else 0) + k
        while (opt.niter if opt.initialized else 0) < target:
            r = opt.run_optimizer(x, 0.01)
            if r["task"] in ("calc_grad", "calc_grad_same_batch"): np.multiply(d, r["requested_on"], out=opt.gradient)
            elif r["task"] == "calc_hess_vec": np.multiply(d, r["requested_on"][1], out=opt.hess_vec)
    advance(120)
    steps = 300 if n <= 100000 else 60
    t0 = time.perf_counter(); advance(steps); dt = time.perf_counter() - t0
    print("host caller SQN m=10 n=%.0e: %.1f us/step" % (n, 1e6 * dt / steps), flush=True)
    opt.release()
