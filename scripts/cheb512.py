import sys, time; sys.path.insert(0,'.')
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
s = polystokes_amd.Solver(0)
for pre, nm in ((abi.PRE_DIAGONAL,'jacobi'),(abi.PRE_CHEBYSHEV,'cheb64'),(abi.PRE_CHEBYSHEV_F32,'cheb32')):
    sc, p = scenes.coil(512, tile=16, pad=2); p.preconditioner = pre
    s.upload(sc, p); s.step_device()
    t0=time.perf_counter(); rc = s.step_device(); ms=(time.perf_counter()-t0)*1e3
    x, b = s.array("solutionVector"), s.array("b")
    res = b - s.apply(x)
    rre = min(res@res, (res@res)/(x@x))
    print(nm, "rc", rc, "iters", int(s.stats.solveData[1]), "ms/step %.1f"%ms, "solve %.1f"%s.stats.stage_ms[8], "true rre %.3e (tol^2 1e-6)"%rre, "cheb32", int(s.array("chebInner32")[0]), "dofs", s.nP+s.nT, flush=True)
print(s.memory_stats())
