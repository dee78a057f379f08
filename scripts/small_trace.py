"""One cavity solve at the given resolution (rocprofv3 --kernel-trace target: kernel durations and gaps in the launch-bound regime)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = polystokes_amd.Solver(0)
sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
s.upload(sc, p); s.step_device(); s.step_device()
print(n, int(s.stats.solveData[1]), float(s.stats.stage_ms[8]), flush=True)
