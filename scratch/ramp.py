"""Per-step wall time of the bench workload right after set-up: do the first steps run slower than the sustained rate
(bench.py's K = 20 .. 40 timed steps read 3 - 5 % below its `sustained` leg)?"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
import stochqn_amd
lib = stochqn_amd.cdll(); be = stochqn_amd.lib()
u64 = C.c_ulonglong
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_synth_uniform.argtypes = [C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double, C.c_double]
lib.stochqn_hip_synth_noisy_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double]
lib.stochqn_hip_synth_batch_row.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, u64, C.c_uint, C.c_uint]
lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = {"lib": lib, "be": be, "dev": dev, "dist": None, "cpu_or_dev": dev, "rank": 0, "world": 1}
wl = bench.Workload(ctx, 100_000_000, 0, 20, 10, 32)
wl.objective()
wl.steps(5)
torch.cuda.synchronize()
ts = []
for i in range(120):
    t0 = time.perf_counter(); wl.one_step(); torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 3))
print(json.dumps({"per_step_ms": ts}))
ordinary = [t for i, t in enumerate(ts) if (i + 6) % 10 not in (0, 1)]
print("ordinary steps: first 10 mean %.3f, steps 10-30 %.3f, last 50 %.3f" % (sum(ordinary[:10]) / 10, sum(ordinary[10:30]) / 20, sum(ordinary[-50:]) / 50))
