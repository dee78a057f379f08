"""How far apart are two correct solves of the same scene?  single domain vs 4 slabs, at two tolerances (AMP check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for tol in (1e-3, 1e-5, 1e-7):
    sc, p = getattr(scenes, name)(n)
    p.preconditioner = abi.PRE_DIAGONAL; p.tolerance = tol; p.maxSolverIterations = 50000
    s = polystokes_amd.Solver(0); rc = s.step(sc, p); it1 = int(s.stats.solveData[1]); v1 = [v.copy() for v in s.vel]; s.close()
    g = polystokes_amd.Group(world); rc2 = g.solve_scene(sc, p); it2 = int(g.stats.solveData[1])
    d = [float(np.abs(g.vel[a] - v1[a]).max() / max(np.abs(v1[a]).max(), 1e-30)) for a in range(3)]
    print(name, n, "tol", tol, "rc", rc, rc2, "iters", it1, it2, "max rel vel diff", ["%.2e" % x for x in d], flush=True)
    g.close()
