"""Per-iteration cost of a scene whose stress diagonal uInv is NOT value-set coded (variable viscosity: > 256 distinct values) against the same scene with a constant viscosity.
usage: vv_penalty.py [res]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
s = polystokes_amd.Solver(0)
for vv in (False, True):
    for pre, nm in ((abi.PRE_DIAGONAL, "jacobi"), (abi.PRE_CHEBYSHEV_F32, "cheb32")):
        sc, p = scenes.blob(n, n, n, seed=5, tile=16, pad=2, variable_viscosity=vv)
        p.preconditioner = pre
        s.upload(sc, p); s.step_device()
        t0 = time.perf_counter(); rc = s.step_device(); ms = (time.perf_counter() - t0) * 1e3
        it = int(s.stats.solveData[1])
        print("blob %d^3 variable_viscosity=%s %s: rc %d, %d DOFs, %d iterations, step %.1f ms, solve %.1f ms = %.1f us per iteration, diagonalsCoded %d, cheb32 %d" % (
            n, vv, nm, rc, s.nP + s.nT, it, ms, s.stats.stage_ms[8], s.stats.stage_ms[8] * 1e3 / max(it, 1), int(s.array("diagonalsCoded")[0]), int(s.array("chebInner32")[0])), flush=True)
