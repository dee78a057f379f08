"""BASELINE configs 2, 4, 5 at their stated sizes on one GPU (ad hoc timing; the -m gpu tests assert the properties).
usage: configs.py <scene> <res> [tol] [world]   (world > 1: in-process group of slabs on one GPU)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
world = int(sys.argv[4]) if len(sys.argv) > 4 else 1
t0 = time.time()
if world == 1:
    sc, p = getattr(scenes, name)(n)
    p.preconditioner = abi.PRE_DIAGONAL; p.tolerance = tol; p.maxSolverIterations = 20000
    print("scene built %.1f s" % (time.time() - t0), flush=True)
    s = polystokes_amd.Solver(0)
    s.upload(sc, p)
    for rep in range(2):
        t0 = time.time(); rc = s.step_device(); dtm = (time.time() - t0) * 1e3
        print(name, n, "rc", rc, "n", s.nP + s.nT, "iters", int(s.stats.solveData[1]), "step ms %.1f" % dtm,
              "setup ms %.1f" % sum(s.stats.stage_ms[i] for i in range(8)), "regions", s.nRegions, "stream runs (S distinct/all, St distinct/all)", list(s.array("streamRuns")), flush=True)
else:
    g = polystokes_amd.Group(world)
    for r in range(world):
        sc, p, sl = scenes.scene_slab(name, n, world, r, precond=abi.PRE_DIAGONAL)
        p.tolerance = tol; p.maxSolverIterations = 20000
        g.ranks[r].upload(sc, p); g.ranks[r].set_slab(sl)
    print("slabs built %.1f s" % (time.time() - t0), flush=True)
    for rep in range(2):
        t0 = time.time(); rc = g.step(); dtm = (time.time() - t0) * 1e3
        print(name, n, "world", world, "rc", rc, "iters", int(g.stats.solveData[1]), "step ms %.1f" % dtm, flush=True)
