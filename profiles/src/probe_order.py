"""Which HIP runtime(s) end up in the process when libstochqn is loaded before torch?  (scratch)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})

def variant(v):
    import ctypes as C
    import numpy as np
    import stochqn_amd
    from harness import NoisyQuadratic, run_trace, OPTIMIZERS
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    be = stochqn_amd.lib()
    print(v, "available:", lib.stochqn_hip_available(), maps(), flush=True)
    P = NoisyQuadratic(300, seed=7)
    if v in ("host_run", "inject"):
        if v == "inject":
            lib.stochqn_hip_set_option(b"fail_alloc_after", 0.0)
        opt = OPTIMIZERS["SQN"](backend=be, space="host", mem_size=3, bfgs_upd_freq=4)
        try:
            run_trace(opt, P, P.x0(), 0.1, 30)
            print(v, "host run ok", flush=True)
        except ValueError as e:
            print(v, "refused:", e, flush=True)
        lib.stochqn_hip_set_option(b"fail_alloc_after", -1.0)
    import torch
    print(v, "torch.cuda.is_available:", torch.cuda.is_available(), maps(), flush=True)
    if torch.cuda.is_available():
        opt = OPTIMIZERS["SQN"](backend=be, space="device", mem_size=3, bfgs_upd_freq=4)
        x = torch.as_tensor(P.x0(), device="cuda")
        run_trace(opt, P, x, 0.1, 30)
        print(v, "device run ok", float(x.sum()), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        variant(sys.argv[1])
    else:
        for v in ("load_only", "host_run", "inject"):
            r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
            print(r.stdout, r.stderr[-1500:], "rc", r.returncode, flush=True)
