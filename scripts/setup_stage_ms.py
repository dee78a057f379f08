import sys, os, json
sys.path.insert(0, os.getcwd())
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
sc, p = scenes.cavity(256, tile=16, pad=2)
p.maxSolverIterations = 2          # the preconditioner stage runs with the solve: two iterations, then the stages are read
s = polystokes_amd.Solver(0); s.upload(sc, p)
for i in range(3):
    s.step_device()
    print({abi.STAGE_NAMES[k]: round(float(s.stats.stage_ms[k]), 2) for k in range(8)}, flush=True)
