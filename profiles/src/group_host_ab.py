"""Host caller (numpy arrays) of the single-process multi-device mode on two virtual shards of one GPU: time inside run_optimizer per
ordinary step with the per-shard overlaps of the host path on (default) and off (upload_slices = 0, apply_chunks = 1, spec_x = 0:
what the group did before its shards took the caller's slices as host pointers).  n = 1e8, m = 20, L = 2 (the ring fills in 40 steps)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochqn_amd
from stochqn_amd import SQN_free
lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
if shards > 1:
    for k, v in ((b"virtual_devices", 1.0), (b"devices_min_n", 1.0), (b"devices", float(shards))):
        assert lib.stochqn_hip_set_option(k, v) == 0
rng = np.random.default_rng(1)
d = 0.5 + rng.random(n)
out = {}
for label, opts in (("overlaps on", {b"upload_slices": 8.0, b"apply_chunks": 8.0, b"spec_x": 1.0}),
                    ("overlaps off", {b"upload_slices": 0.0, b"apply_chunks": 1.0, b"spec_x": 0.0}),
                    ("overlaps on again", {b"upload_slices": 8.0, b"apply_chunks": 8.0, b"spec_x": 1.0})):
    for k, v in opts.items():
        assert lib.stochqn_hip_set_option(k, v) == 0
    opt = SQN_free(mem_size=20, bfgs_upd_freq=2, min_curvature=None, backend=stochqn_amd.lib(), space="host")
    x = 1.0 + rng.random(n)
    times = []
    for call in range(130):
        t0 = time.perf_counter()
        r = opt.run_optimizer(x, 0.05)
        dt = time.perf_counter() - t0
        task = r["task"]
        if task == "calc_hess_vec":
            rx, rv = r["requested_on"]
            opt.update_hess_vec(d * np.asarray(rv))
        else:
            opt.update_gradient(d * np.asarray(r["requested_on"]))
        if opt.BFGS_mem.mem_used == 20 and task == "calc_grad" and r["info"]["x_changed_in_run"]:
            times.append(dt)
    times.sort()
    out[label] = {"ordinary_steps_timed": len(times), "fastest_ms": round(1e3 * times[0], 2), "median_ms": round(1e3 * times[len(times) // 2], 2)}
    print(json.dumps({"shards": shards, "n": n, "mode": label, **out[label]}), flush=True)
    opt.release()
    lib.stochqn_hip_release_all()
