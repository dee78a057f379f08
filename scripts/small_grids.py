"""us per CG iteration on small cavities (launch-bound regime) and the per-kernel times at 128^3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
s = polystokes_amd.Solver(0)
sizes = tuple(int(a) for a in sys.argv[1:]) or (32, 64, 96, 128)
for n in sizes:
    sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    s.upload(sc, p); s.step_device()
    best = 1e30
    for _ in range(3):
        s.step_device(); best = min(best, float(s.stats.stage_ms[8]))
    it = int(s.stats.solveData[1])
    print("cavity %d^3: n = %d DOFs, %d iterations, solve %.2f ms -> %.1f us per iteration (%d launches, cache policy level %s)" % (n, s.nP + s.nT, it, best, best * 1e3 / max(it, 1), 4 if int(s.array("fusedStep")[0]) else 5, os.environ.get("PS_NT_LEVEL", "auto")), flush=True)
