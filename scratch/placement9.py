"""Do S, Y and g cut from ONE allocation always land in the same state?  Six fresh 34 GB blocks (S | Y | g, then g | S | Y), each measured;
between them spacers of growing size stay allocated so that every block lands somewhere else.  Then, for comparison, the same six times
with three separate allocations."""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stochqn_amd
lib = stochqn_amd.cdll()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
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

def kernels():
    out = {}
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value: out[lib.stochqn_hip_profile_name(i).decode()] = round(ms.value / cnt.value, 4)
    return out

def alloc(bytes_):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), bytes_) == 0
    return p.value

def fill(S, Y):
    for k in range(m):
        lib.stochqn_hip_synth_uniform(S + 8 * k * n, n, 0, 1, 1, k, -0.5e-3, 1e-3)
        lib.stochqn_hip_synth_noisy_grad(Y + 8 * k * n, d.data_ptr(), S + 8 * k * n, n, 0, 1, 9, 0, 0.0)
    torch.cuda.synchronize()

def measure(S, Y, g, reps=10):
    lib.stochqn_hip_set_option(b"raw_reuse_cache", 1.0)
    for _ in range(2):
        hip.hipMemcpy(g, g0.data_ptr(), 8 * n, 3); assert lib.stochqn_hip_two_loop(g, n, None, 0.0, Y, S, m, m, 3, rho.ctypes.data, alpha.ctypes.data) == 0
    lib.stochqn_hip_profile_enable(1); lib.stochqn_hip_profile_reset()
    for _ in range(reps):
        hip.hipMemcpy(g, g0.data_ptr(), 8 * n, 3); lib.stochqn_hip_two_loop(g, n, None, 0.0, Y, S, m, m, 3, rho.ctypes.data, alpha.ctypes.data)
    torch.cuda.synchronize(); lib.stochqn_hip_profile_enable(0)
    k = kernels()
    lib.stochqn_hip_release_all()
    return {x: k.get(x) for x in ("sdot", "qdot", "sadd")}

lib.stochqn_hip_set_option(b"twopass_kappa_max", 0.0)
MB = 1 << 20
held = []
def fill_block(S, Y):
    lib.stochqn_hip_synth_uniform(S, m * n, 0, 1, 1, 0, -0.5e-3, 1e-3)
    lib.stochqn_hip_synth_uniform(Y, m * n, 0, 1, 2, 0, -0.5e-3, 1e-3)
    torch.cuda.synchronize()
for rep in range(6):
    block = alloc(2 * 8 * m * n + 8 * n + 64 * MB)
    if rep % 2 == 0: S, Y, g = block, block + 8 * m * n, block + 16 * m * n
    else:            g, S, Y = block, block + 8 * n + 2 * MB, block + 8 * n + 2 * MB + 8 * m * n
    fill_block(S, Y)
    print(json.dumps({"layout": "one block, " + ("S|Y|g" if rep % 2 == 0 else "g|S|Y"), "rep": rep, "block": hex(block), **measure(S, Y, g, reps=6)}), flush=True)
    hip.hipFree(block)
    held.append(alloc((rep + 1) * 1_077_594_624))
for rep in range(6):
    S, Y, g = alloc(8 * m * n), alloc(8 * m * n), alloc(8 * n)
    fill_block(S, Y)
    print(json.dumps({"layout": "three allocations", "rep": rep, "S": hex(S), "Y": hex(Y), "g": hex(g), **measure(S, Y, g, reps=6)}), flush=True)
    hip.hipFree(S); hip.hipFree(Y); hip.hipFree(g)
    held.append(alloc((rep + 1) * 577_594_624))
