"""ms/step and iterations of the three preconditioners (identity, Jacobi, Chebyshev degree k) on one scene/size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
s = polystokes_amd.Solver(0)
for pre, deg in ((abi.PRE_IDENTITY, 0), (abi.PRE_DIAGONAL, 0), (abi.PRE_CHEBYSHEV, 2), (abi.PRE_CHEBYSHEV, 3), (abi.PRE_CHEBYSHEV, 4), (abi.PRE_CHEBYSHEV, 5), (abi.PRE_CHEBYSHEV, 6), (abi.PRE_CHEBYSHEV, 8)):
    sc, p = getattr(scenes, name)(n)
    p.preconditioner, p.preconditionerDegree = pre, deg
    s.upload(sc, p)
    s.step_device()
    t0 = time.perf_counter(); rc = s.step_device(); ms = (time.perf_counter() - t0) * 1e3
    print(name, n, "pre", pre, "deg", deg, "rc", rc, "iters", int(s.stats.solveData[1]), "step ms %.1f" % ms, "solve ms %.1f" % s.stats.stage_ms[8], "precond setup ms %.1f" % s.stats.stage_ms[7], flush=True)
