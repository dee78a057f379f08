"""One rank, torch imported and its GPU context live (as in bench.py), then the library's RCCL communicator (world 1):
init, self-test, a slab step, clean exit.  Shows which librccl the process ends up with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
_ = torch.zeros(4, device="cuda")
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
maps = [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l]
print("rccl mapped before:", sorted(set(maps)), flush=True)
sc, p, slab = scenes.cavity_slab(32, 1, 0, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.set_slab(slab)
s.comm_init(polystokes_amd.comm_unique_id(), 0, 1)
s.comm_selftest()
rc = s.step_device()
maps = [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l]
print("rccl mapped after:", sorted(set(maps)), "rc", rc, "iterations", int(s.stats.solveData[1]), flush=True)
s.close()
print("closed", flush=True)
