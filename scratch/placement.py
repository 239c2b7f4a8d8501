"""Where does the run-to-run spread of pass 2 / pass 3 come from (2.85 .. 3.27 ms, pass 1 unaffected)?  Same process, the
three-pass two-loop on the same data: (a) repeated as is, (b) with g moved to other addresses, (c) with S and Y moved."""
import ctypes as C, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stochqn_amd
lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_profile_name.restype = C.c_char_p
lib.stochqn_hip_two_loop.restype = C.c_int
lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
n, m = 100_000_000, 20
rho, alpha = np.zeros(m), np.zeros(m)

def kernels():
    out = {}
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value: out[lib.stochqn_hip_profile_name(i).decode()] = round(ms.value / cnt.value, 4)
    return out

def fill(S, Y):
    g = torch.Generator(device=dev).manual_seed(1)
    d = 0.5 + torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    for k in range(m):
        S[k * n:(k + 1) * n] = 1e-3 * (torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5)
        torch.mul(d, S[k * n:(k + 1) * n], out=Y[k * n:(k + 1) * n])

def measure(S, Y, g0, gq, reps=12):
    lib.stochqn_hip_set_option(b"raw_reuse_cache", 1.0)
    for _ in range(2):
        gq.copy_(g0); lib.stochqn_hip_two_loop(gq.data_ptr(), n, None, 0.0, Y.data_ptr(), S.data_ptr(), m, m, 3, rho.ctypes.data, alpha.ctypes.data)
    lib.stochqn_hip_profile_enable(1); lib.stochqn_hip_profile_reset()
    for _ in range(reps):
        gq.copy_(g0); lib.stochqn_hip_two_loop(gq.data_ptr(), n, None, 0.0, Y.data_ptr(), S.data_ptr(), m, m, 3, rho.ctypes.data, alpha.ctypes.data)
    torch.cuda.synchronize(); lib.stochqn_hip_profile_enable(0)
    k = kernels()
    return {x: k.get(x) for x in ("sdot", "qdot", "sadd")}

S = torch.empty(m * n, dtype=torch.float64, device=dev); Y = torch.empty(m * n, dtype=torch.float64, device=dev)
fill(S, Y)
g0 = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
gq = torch.empty_like(g0)
print("addresses S %x Y %x gq %x" % (S.data_ptr(), Y.data_ptr(), gq.data_ptr()), flush=True)
for rep in range(4):
    print(json.dumps({"case": "same arrays", "rep": rep, **measure(S, Y, g0, gq)}), flush=True)
# (b) g at other addresses: sub-views of a bigger buffer at various byte offsets
big = torch.empty(n + (1 << 24), dtype=torch.float64, device=dev)
for off in (0, 512, 8192, 1 << 15, 1 << 17, (1 << 18) + 512, 1 << 20, (1 << 21) + (1 << 12), 1 << 23):
    v = big[off:off + n]
    lib.stochqn_hip_release_all()
    print(json.dumps({"case": "g at buffer + %d doubles" % off, "addr": hex(v.data_ptr()), **measure(S, Y, g0, v)}), flush=True)
del big
# (c) fresh allocations of everything, three times (different physical pages; torch's cache emptied in between)
for rep in range(3):
    lib.stochqn_hip_release_all()
    del S, Y, gq
    torch.cuda.empty_cache()
    pad = torch.empty((rep + 1) * 123_457_000, dtype=torch.uint8, device=dev)
    S = torch.empty(m * n, dtype=torch.float64, device=dev); Y = torch.empty(m * n, dtype=torch.float64, device=dev)
    gq = torch.empty(n, dtype=torch.float64, device=dev)
    fill(S, Y)
    print(json.dumps({"case": "fresh allocations", "rep": rep, "S": hex(S.data_ptr()), "Y": hex(Y.data_ptr()), "g": hex(gq.data_ptr()), **measure(S, Y, g0, gq)}), flush=True)
    del pad
# (d) stability over time: 40 s of back-to-back calls
t0 = time.time()
while time.time() - t0 < 40:
    r = measure(S, Y, g0, gq, reps=40)
    print(json.dumps({"case": "sustained", "t": round(time.time() - t0, 1), **r}), flush=True)
