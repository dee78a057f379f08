import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import polystokes_amd
from polystokes_amd import _abi as abi
from helpers import fuzz_brick_case
seeds = [int(v) for v in sys.argv[1:]] or [4237, 4226, 4211, 4230]     # usage: fuzz_bricks_seeds_tol.py [seed ...]
for seed in seeds:
    for tol in ([float(v) for v in os.environ["TOLS"].split(",")] if os.environ.get("TOLS") else (1e-6, 1e-9)):
        sc, p, dims, n, tile = fuzz_brick_case(seed, tol); p.maxSolverIterations = 100000
        s = polystokes_amd.Solver(0); rc1 = s.step(sc, p); it1 = int(s.stats.solveData[1])
        g = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims); rc2 = g.solve_scene(sc, p); it2 = int(g.stats.solveData[1])
        d = max(float(np.abs(g.vel[a] - s.vel[a]).max() / max(np.abs(s.vel[a]).max(), 1e-30)) for a in range(3))
        print(seed, dims, "tol %.0e" % tol, "rc", rc1, rc2, "iters", it1, it2, "bicg", s.stats.usedBiCGStab, g.stats.usedBiCGStab, "vel diff %.2e" % d, flush=True)
        g.close(); s.close()
