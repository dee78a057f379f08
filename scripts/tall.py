"""iteration count / time of the weak-scaling global problems (n x n x n*w cavity) solved on ONE GPU"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]); ws = [int(w) for w in sys.argv[2].split(",")]
for w in ws:
    sc0, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    velx = np.zeros((n * w, n, n + 1), np.float32); velx[n * w - 1] = 1.0
    sc = abi.Scene(n, n, n * w, sc0.dx, sc0.dt, 1.0, [velx, 0.0, 0.0], np.float32(-1.0), np.float32(1.0), 1.0)
    s = polystokes_amd.Solver(0); s.upload(sc, p)
    t0 = time.time(); rc = s.step_device(); dt = time.time() - t0
    print("w", w, "rc", rc, "iters", int(s.stats.solveData[1]), "dofs", s.nP + s.nT, "ms %.1f" % (dt * 1e3), flush=True)
    s.close()
