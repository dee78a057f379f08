"""Chebyshev interval ratio / degree sweep (PS_CHEB_RATIO is read once per process: one ratio per run)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
s = polystokes_amd.Solver(0)
for deg in (3, 4, 5, 6):
    sc, p = getattr(scenes, name)(n)
    p.preconditioner, p.preconditionerDegree = abi.PRE_CHEBYSHEV, deg
    s.upload(sc, p); s.step_device(); s.step_device()
    print(name, n, "ratio", os.environ.get("PS_CHEB_RATIO", "30"), "deg", deg, "iters", int(s.stats.solveData[1]), "solve ms %.1f" % s.stats.stage_ms[8], flush=True)
