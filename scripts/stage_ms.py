"""Stage times (ms) of one warm step: usage stage_ms.py <scene> <res>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
sc, p = getattr(scenes, name)(n)
p.preconditioner = abi.PRE_DIAGONAL
s = polystokes_amd.Solver(0)
s.upload(sc, p); s.step_device(); s.step_device()
print(name, n, "iterations", int(s.stats.solveData[1]), {abi.STAGE_NAMES[i]: round(float(s.stats.stage_ms[i]), 2) for i in range(len(abi.STAGE_NAMES))}, flush=True)
