"""Degree sweep of the Chebyshev polynomial with fp32 inner vectors (PS_PRE_CHEBYSHEV_F32) against the fp64 form: ms/step and iterations on one scene.
usage: cheb32_degree.py [scene] [res]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = (sys.argv[1] if len(sys.argv) > 1 else "cavity"), int(sys.argv[2]) if len(sys.argv) > 2 else 256
s = polystokes_amd.Solver(0)
for pre in (abi.PRE_CHEBYSHEV_F32, abi.PRE_CHEBYSHEV):
    for deg in (3, 4, 5, 6, 8, 10, 12):
        sc, p = getattr(scenes, name)(n, tile=16, pad=2)
        p.preconditioner, p.preconditionerDegree = pre, deg
        s.upload(sc, p)
        s.step_device()
        t0 = time.perf_counter(); rc = s.step_device(); ms = (time.perf_counter() - t0) * 1e3
        print(name, n, "fp32" if pre == abi.PRE_CHEBYSHEV_F32 else "fp64", "degree", deg, "rc", rc, "iters", int(s.stats.solveData[1]), "step ms %.1f" % ms, "solve ms %.1f" % s.stats.stage_ms[8], flush=True)
