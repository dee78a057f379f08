"""us per CG iteration of the coil scene (BASELINE config 2 at 128^3) under the environment's grid switches.  usage: coil_grids.py [res]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = polystokes_amd.Solver(0)
sc, p = scenes.coil(n)
p.preconditioner = abi.PRE_DIAGONAL
s.upload(sc, p); s.step_device()
best = 1e30
for _ in range(5):
    s.step_device(); best = min(best, float(s.stats.stage_ms[8]))
it = int(s.stats.solveData[1])
env = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("PS_") and k not in ("PS_LIB",))
print("%-50s coil %d^3: %d DOFs, %d iterations, solve %.3f ms -> %.1f us per iteration (%d launches)" % (env or "(default)", n, s.nP + s.nT, it, best, best * 1e3 / max(it, 1), 4 if int(s.array("fusedStep")[0]) else 5), flush=True)
