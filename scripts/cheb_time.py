"""Chebyshev-PCG and five-kernel Jacobi-PCG solve times at 256^3 (grid-size A/B of the St kernel's other modes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
s = polystokes_amd.Solver(0)
for pre in (abi.PRE_CHEBYSHEV, abi.PRE_DIAGONAL):
    sc, p = scenes.cavity(256, precond=pre)
    s.upload(sc, p); s.step_device(); s.step_device()
    print("precond", pre, "iterations", int(s.stats.solveData[1]), "solve ms %.1f" % s.stats.stage_ms[8], "fused", int(s.array("fusedStep")[0]), flush=True)
