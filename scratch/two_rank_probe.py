"""Probe: can two ranks share the one GPU of the test box (for rehearsing the N>1 path)?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import stochqn_amd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
lib = stochqn_amd.cdll()
uid = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    buf = (C.c_ubyte * 128)(); assert lib.stochqn_hip_comm_unique_id(buf) == 0
    uid = torch.tensor(list(buf), dtype=torch.uint8)
dist.broadcast(uid, 0)
rc = lib.stochqn_hip_comm_init(rank, world, bytes(uid.tolist()))
print("rank", rank, "comm_init rc", rc, flush=True)
dist.destroy_process_group()
