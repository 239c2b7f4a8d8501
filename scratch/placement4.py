"""Follow-up of placement3 (the state follows g, the vector that is read AND written): does a physically contiguous g, or a
g cut from a power-of-two allocation, always land in the fast state?  Fixed S and Y."""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "placement3.py")).read().split("S, Y = alloc(8 * m * n), alloc(8 * m * n)")[0])
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]

def alloc_contig(bytes_):
    p = C.c_void_p()
    assert hip.hipExtMallocWithFlags(C.byref(p), bytes_, 0x4) == 0
    return p.value

S, Y = alloc(8 * m * n), alloc(8 * m * n)
fill(S, Y)
held = []
for rep in range(8):
    for kind, fn in (("hipMalloc", lambda: alloc(8 * n)), ("contiguous", lambda: alloc_contig(8 * n)), ("1 GiB block", lambda: alloc(1 << 30))):
        g = fn()
        print(json.dumps({"g": kind, "rep": rep, "addr": hex(g), **measure(S, Y, g)}), flush=True)
        held.append(g)
    held.append(alloc((rep + 1) * 77_594_624))
