import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from harness import *
from oracle import oracle
from test_gpu_parity import CONFIGS
import stochqn_amd
cfg = [c for c in CONFIGS if c[0]=="adaqn_fisher_adagrad_nomaxincr"][0]
name, optname, kw, step, calls, pkw = cfg
for n in (4097, 1000):
    P = NoisyQuadratic(n, seed=7, **pkw)
    want = run_trace(OPTIMIZERS[optname](backend=oracle.bound(), **kw), P, P.x0(), step, calls)
    x = torch.as_tensor(P.x0(), device="cuda")
    got = run_trace(OPTIMIZERS[optname](space="device", **kw), P, x, step, calls)
    print(n, ["%.1e"%rel_err(g["x"], w["x"]) for g,w in zip(got,want)])
