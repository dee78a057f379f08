"""AMP spread at full size (spheres 256^3, Jacobi): single domain vs 4 slabs, per tolerance, for the kernel variants given by the
environment (PS_S_DUAL / PS_ST_DUAL) — how much of the velocity difference between two valid solves is the rounding path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
tols = [float(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1e-7]
for tol in tols:
    sc, p = scenes.spheres(256)
    p.preconditioner = abi.PRE_DIAGONAL; p.tolerance = tol; p.maxSolverIterations = 100000
    s = polystokes_amd.Solver(0); rc = s.step(sc, p); it1 = int(s.stats.solveData[1]); v1 = [v.copy() for v in s.vel]; s.close()
    g = polystokes_amd.Group(4); rc2 = g.solve_scene(sc, p); it2 = int(g.stats.solveData[1])
    d = max(float(np.abs(g.vel[a] - v1[a]).max() / max(np.abs(v1[a]).max(), 1e-30)) for a in range(3))
    print("S_DUAL=%s ST_DUAL=%s tol %.0e iters %d / %d  max rel velocity difference %.2e" % (os.environ.get("PS_S_DUAL", "1"), os.environ.get("PS_ST_DUAL", "1"), tol, it1, it2, d), flush=True)
    g.close()
