"""One Chebyshev-PCG solve of the 256^3 cavity (rocprofv3 --kernel-trace --stats target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
s = polystokes_amd.Solver(0)
sc, p = scenes.cavity(256, precond=abi.PRE_CHEBYSHEV)
s.upload(sc, p); s.step_device(); s.step_device()
print(256, int(s.stats.solveData[1]), float(s.stats.stage_ms[8]), flush=True)
