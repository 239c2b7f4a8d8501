import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from harness import *
from oracle import oracle
import stochqn_amd
P = Rosenbrock2D()
for space in ("device","host"):
    want = run_trace(oLBFGS_free(mem_size=5, backend=oracle.bound()), P, P.x0(), 0.1, 12)
    x = P.x0()
    if space=="device": x = torch.as_tensor(x, device="cuda")
    got = run_trace(oLBFGS_free(mem_size=5, space=space), P, x, 0.1, 12)
    for i,(g,w) in enumerate(zip(got,want)):
        print(space, i, g["task"], g["info"], g["x"], g["mem_used"], g["mem_st_ix"], "| want", w["task"], w["info"], w["x"], w["mem_used"], w["mem_st_ix"])
