"""Does a physically contiguous allocation (hipExtMallocWithFlags, hipDeviceMallocContiguous) of S and Y take the luck out
of the placement?  Several fresh allocations of each kind in one process, the three-pass two-loop timed per kernel."""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stochqn_amd
lib = stochqn_amd.cdll()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_profile_name.restype = C.c_char_p
lib.stochqn_hip_two_loop.restype = C.c_int
lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
u64 = C.c_ulonglong
lib.stochqn_hip_synth_uniform.argtypes = [C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double, C.c_double]
lib.stochqn_hip_synth_noisy_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double]
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
n, m = 100_000_000, 20
rho, alpha = np.zeros(m), np.zeros(m)
d = torch.empty(n, dtype=torch.float64, device=dev)
lib.stochqn_hip_synth_uniform(d.data_ptr(), n, 0, 1, 0, 0, 0.5, 1.0)
g0 = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
gq = torch.empty_like(g0)

def kernels():
    out = {}
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value: out[lib.stochqn_hip_profile_name(i).decode()] = round(ms.value / cnt.value, 4)
    return out

def alloc(flags):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), m * n * 8) if flags is None else hip.hipExtMallocWithFlags(C.byref(p), m * n * 8, flags)
    assert rc == 0, rc
    return p.value

def fill(S, Y):
    for k in range(m):
        lib.stochqn_hip_synth_uniform(S + 8 * k * n, n, 0, 1, 1, k, -0.5e-3, 1e-3)
        lib.stochqn_hip_synth_noisy_grad(Y + 8 * k * n, d.data_ptr(), S + 8 * k * n, n, 0, 1, 9, 0, 0.0)
    torch.cuda.synchronize()

def measure(S, Y, reps=12):
    lib.stochqn_hip_set_option(b"raw_reuse_cache", 1.0)
    for _ in range(2):
        gq.copy_(g0); assert lib.stochqn_hip_two_loop(gq.data_ptr(), n, None, 0.0, Y, S, m, m, 3, rho.ctypes.data, alpha.ctypes.data) == 0
    lib.stochqn_hip_profile_enable(1); lib.stochqn_hip_profile_reset()
    for _ in range(reps):
        gq.copy_(g0); lib.stochqn_hip_two_loop(gq.data_ptr(), n, None, 0.0, Y, S, m, m, 3, rho.ctypes.data, alpha.ctypes.data)
    torch.cuda.synchronize(); lib.stochqn_hip_profile_enable(0)
    k = kernels()
    lib.stochqn_hip_release_all()
    return {x: k.get(x) for x in ("sdot", "qdot", "sadd")}

pads = []
for rep in range(5):
    for name, flags in (("hipMalloc", None), ("contiguous", 0x4)):
        try:
            S, Y = alloc(flags), alloc(flags)
        except AssertionError as e:
            print(json.dumps({"kind": name, "rep": rep, "error": str(e)}), flush=True)
            continue
        fill(S, Y)
        print(json.dumps({"kind": name, "rep": rep, "S": hex(S), "Y": hex(Y), **measure(S, Y)}), flush=True)
        hip.hipFree(S); hip.hipFree(Y)
    p = C.c_void_p(); hip.hipMalloc(C.byref(p), (rep + 1) * 301_989_888 + 4096 * rep); pads.append(p)     # shift what comes next
