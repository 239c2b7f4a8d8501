"""Two-loop time versus the number of pairs in use (raw entry point, n = 1e8, fp64); scratch."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stochqn_amd
lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_profile_name.restype = C.c_char_p
lib.stochqn_hip_two_loop.restype = C.c_int
lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
m = 20
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
d = 0.5 + torch.rand(n, dtype=torch.float64, device=dev, generator=gen)
S = torch.empty(m * n, dtype=torch.float64, device=dev); Y = torch.empty(m * n, dtype=torch.float64, device=dev)
for k in range(m):
    s = 1e-3 * (torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5)
    S[k * n:(k + 1) * n] = s; Y[k * n:(k + 1) * n] = d * s
g0 = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
rho, alpha = np.zeros(m), np.zeros(m)
def kernels():
    out = {}
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value: out[lib.stochqn_hip_profile_name(i).decode()] = round(ms.value / cnt.value, 3)
    return out
for form in (1, 0):
    lib.stochqn_hip_set_option(b"twopass", float(form))
    for used in (1, 2, 3, 4, 6, 8, 10, 12, 16, 20):
        g = g0.clone()
        for rep in range(3):
            g.copy_(g0); lib.stochqn_hip_two_loop(g.data_ptr(), n, None, 0.0, Y.data_ptr(), S.data_ptr(), m, used, 3, rho.ctypes.data, alpha.ctypes.data)
        torch.cuda.synchronize(); lib.stochqn_hip_profile_enable(1); lib.stochqn_hip_profile_reset()
        ts = []
        for rep in range(5):
            g.copy_(g0); torch.cuda.synchronize(); t0 = time.perf_counter()
            lib.stochqn_hip_two_loop(g.data_ptr(), n, None, 0.0, Y.data_ptr(), S.data_ptr(), m, used, 3, rho.ctypes.data, alpha.ctypes.data)
            ts.append(time.perf_counter() - t0)
        lib.stochqn_hip_profile_enable(0)
        t = sorted(ts)[len(ts) // 2]
        words = (4 * used + 3) if form else 8 * used
        print("form %s used %2d: %.3f ms  %.0f GB/s on bytes moved  %s" % ("twopass" if form else "sweeps ", used, 1e3 * t, words * n * 8 / t / 1e9, kernels()), flush=True)
