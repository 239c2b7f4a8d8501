"""Per-step wall time versus n: libstochqn (device-resident C caller, tools/latency.hip) next to the CPU
oracle on this box's host cores.  Run on the GPU box:  python tools/latency_table.py > gpurun_out/latency.log

The CPU column is the oracle (test infrastructure) used as the timed baseline, exactly as bench.py's
cpu_baseline does; nothing here is on the product path."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    exe = os.path.join(ROOT, "tools", "latency")
    src = os.path.join(ROOT, "tools", "latency.hip")
    libdir = os.path.join(ROOT, "stochqn_amd", "lib")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"), src,
                               "-L", libdir, "-lstochqn", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def gpu_us(exe, kind, n, m, steps):
    out = subprocess.check_output([exe, kind, str(n), str(m), str(steps), "0"]).decode()
    return float(re.search(r"([\d.]+) us/step", out).group(1))


def cpu_us(kind, n, m, steps, threads):
    import numpy as np
    from harness import OPTIMIZERS
    from oracle import oracle
    be = oracle.bound()
    oracle.set_threads(threads)
    rng = np.random.default_rng(1)
    d = 0.5 + rng.random(n)
    x = 1.0 + rng.random(n)
    if kind == "sqn":
        opt = OPTIMIZERS["SQN"](backend=be, space="host", mem_size=m, bfgs_upd_freq=10, min_curvature=None)
    else:
        opt = OPTIMIZERS["oLBFGS"](backend=be, space="host", mem_size=m, min_curvature=None)

    def advance(k):
        target = (opt.niter if opt.initialized else 0) + k
        while (opt.niter if opt.initialized else 0) < target:
            r = opt.run_optimizer(x, 0.01)
            if r["task"] in ("calc_grad", "calc_grad_same_batch"):
                np.multiply(d, r["requested_on"], out=opt.gradient)
            elif r["task"] == "calc_hess_vec":
                np.multiply(d, r["requested_on"][1], out=opt.hess_vec)

    advance(3 * m + 25)
    t0 = time.perf_counter()
    advance(steps)
    return 1e6 * (time.perf_counter() - t0) / steps


if __name__ == "__main__":
    exe = build()
    m = 10
    from oracle import oracle as _o
    print("cpu_count:", os.cpu_count(), "usable (affinity, cgroup quota):", _o.usable_cpus(), flush=True)
    print("| optimiser | n | MI355X us/step | CPU oracle us/step (best of 1 / 4 / all usable threads; ctypes caller) | threads | ratio |", flush=True)
    print("|---|---|---|---|---|---|", flush=True)
    for kind in ("olbfgs", "sqn"):
        for n in (1000, 10000, 100000, 1000000, 10000000):
            g = gpu_us(exe, kind, n, m, 500 if n < 10**7 else 200)
            steps = 400 if n <= 10**5 else (60 if n == 10**6 else 12)
            from oracle import oracle
            c, th = min((cpu_us(kind, n, m, steps, t), t) for t in sorted({1, 4, oracle.usable_cpus()}))
            print("| %s m=%d | %.0e | %.1f | %.1f | %d | %.2f |" % (kind, m, n, g, c, th, c / g), flush=True)
