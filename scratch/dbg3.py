import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, ctypes as C
from harness import *
from oracle import oracle
from test_gpu_parity import CONFIGS
import stochqn_amd
cfg=[c for c in CONFIGS if c[0]=="adaqn_nonan_check"][0]
name,optname,kw,step,calls,pkw=cfg
for rep in range(3):
    P=NoisyQuadratic(1000,seed=7,**pkw)
    want=run_trace(OPTIMIZERS[optname](backend=oracle.bound(),**kw),P,P.x0(),step,8)
    opt=OPTIMIZERS[optname](space="host",**kw)
    x=P.x0(); got=run_trace(opt,P,x,step,8)
    for i,(g,w) in enumerate(zip(got,want)):
        print(rep,i,g["task"],g["info"],"err",rel_err(g["x"],w["x"]), "nan in x:", np.isnan(g["x"]).sum(), g["mem_used"])
