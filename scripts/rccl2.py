"""2 processes, ONE GPU: try the raw-RCCL distributed path (may be refused by RCCL as 'duplicate GPU')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import polystokes_amd
from polystokes_amd import scenes, partition, _abi as abi
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
sc, p, slab = scenes.cavity_slab(32, world, rank, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
obj = [polystokes_amd.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(obj, 0)
s.set_slab(slab)
s.comm_init(obj[0], rank, world)
rc = s.step_device()
print("rank", rank, "rc", rc, "iters", s.stats.solveData[1], flush=True)
dist.barrier()
