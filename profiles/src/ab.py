"""In-process A/B of library options on the two-pass SQN step (same box, interleaved)."""
import ctypes as C, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stochqn_amd
from stochqn_amd import SQN_free, oLBFGS_free, adaQN_free
lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_profile_name.restype = C.c_char_p
dev = torch.device("cuda", 0)

def kernels():
    out = {}
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value: out[lib.stochqn_hip_profile_name(i).decode()] = round(ms.value / cnt.value, 3)
    return out

def setup(kind, n, m, L):
    g = torch.Generator(device=dev).manual_seed(1)
    d = 0.5 + torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    dn = d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 1))
    x = 1 + torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    if kind in ("sqn", "sqn0"):
        opt = SQN_free(mem_size=m, bfgs_upd_freq=1, min_curvature=None, space="device", check_nan=(kind == "sqn"))
    elif kind == "adaqn":
        opt = adaQN_free(mem_size=m, fisher_size=16, bfgs_upd_freq=2, max_incr=None, min_curvature=None, scal_reg=1e-4,
                         rmsprop_weight=0.9, space="device")
    else:
        opt = oLBFGS_free(mem_size=m, min_curvature=None, space="device")
    return opt, d, dn, x

INFOS = {}
def advance(opt, d, dn, x, step, k):
    target = (opt.niter if opt.initialized else 0) + k
    while (opt.niter if opt.initialized else 0) < target:
        r = opt.run_optimizer(x, step)
        INFOS[r["info"]["iteration_info"]] = INFOS.get(r["info"]["iteration_info"], 0) + 1
        if r["task"] in ("calc_grad", "calc_grad_same_batch"): torch.mul(dn, r["requested_on"], out=opt.gradient)
        elif r["task"] == "calc_hess_vec": torch.mul(d, r["requested_on"][1], out=opt.hess_vec)

def measure(opt, d, dn, x, step, steps, opts):
    for k, v in opts.items(): lib.stochqn_hip_set_option(k.encode(), float(v))
    advance(opt, d, dn, x, step, 3)
    torch.cuda.synchronize(); lib.stochqn_hip_profile_enable(1); lib.stochqn_hip_profile_reset()
    t0 = time.perf_counter(); advance(opt, d, dn, x, step, steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0; lib.stochqn_hip_profile_enable(0)
    return round(1e3 * dt / steps, 3), kernels()

if __name__ == "__main__":
    kind, n, m = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3])
    variants = [json.loads(a) for a in sys.argv[4:]] or [{}]
    opt, d, dn, x = setup(kind, n, m, 1)
    advance(opt, d, dn, x, 1e-4 if kind == "adaqn" else 0.01, (2 * m + 6) if kind == "adaqn" else (m + 3))   # fill the ring
    print("after fill: niter", opt.niter, "mem_used", opt.BFGS_mem.mem_used, INFOS, flush=True)
    if kind == "adaqn":
        opt.BFGS_mem.upd_freq = opt.bfgs_upd_freq = 1000000      # no more pair updates: time the step path only
    if kind in ("sqn", "sqn0"):
        opt.BFGS_mem.upd_freq = opt.bfgs_upd_freq = 10
        opt.niter = 10 * ((opt.niter + 9) // 10)
    base = {"grid_cap": 0, "rows_grid": 0, "rows_split": 0, "reverse": 1, "twopass": 1, "nontemporal": 1, "combine_batch": 8, "twopass_h0": 1, "h0_per_cu": 0, "rows_waves": 0, "fold_coef": 1, "keep_tail": 0, "fuse_apply": 1}
    for rep in range(2):
        for v in variants:
            o = dict(base); o.update(v)
            ms, k = measure(opt, d, dn, x, 1e-4 if kind == "adaqn" else 0.01, 20, o)
            print(json.dumps({"variant": v, "ms_per_step": ms, "kernels": k}), flush=True)
