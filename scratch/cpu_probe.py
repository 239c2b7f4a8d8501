import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip(), flush=True)
    except Exception as e: print(f, "n/a", flush=True)
from oracle import oracle
n, m = 10**7, 20
rng = np.random.default_rng(0)
S = 1e-3 * (rng.random(m * n) - 0.5); d = 0.5 + rng.random(n)
Y = (S.reshape(m, n) * d).reshape(-1)
for th in (1, 4, 8, 16, 32, 64, 128):
    oracle.set_threads(th)
    g = rng.random(n) - 0.5
    oracle.two_loop(g, None, 0.0, Y, S, m, m, 3)
    t0 = time.perf_counter()
    for _ in range(2): oracle.two_loop(g, None, 0.0, Y, S, m, m, 3)
    print("threads", th, "two-loop n=1e7 m=20: %.1f ms" % (1e3 * (time.perf_counter() - t0) / 2), flush=True)
