import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = polystokes_amd.Solver(0)
for tol in (1.3e-3, 1.2e-3, 1.1e-3, 1.05e-3, 1.02e-3, 1.0e-3, 0.98e-3, 0.95e-3, 0.9e-3, 0.8e-3, 0.7e-3):
    sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    p.tolerance = tol
    rc = s.step(sc, p)
    print(n, "tol %.3g" % tol, "iters", int(s.stats.solveData[1]), "err %.4g" % s.stats.solveData[0], flush=True)
