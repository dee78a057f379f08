import sys, os, collections; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pre = {"jacobi": abi.PRE_DIAGONAL, "identity": abi.PRE_IDENTITY, "chebyshev": abi.PRE_CHEBYSHEV}[sys.argv[3] if len(sys.argv) > 3 else "jacobi"]
sc, p = scenes.cavity(n, tile=16, pad=2, precond=pre)
s = polystokes_amd.Solver(0); s.upload(sc, p)
hist = collections.Counter()
for i in range(reps):
    rc = s.step_device()
    import hashlib
    hist[(rc, int(s.stats.solveData[1]), float(s.stats.solveData[0]).hex(), hashlib.sha1(s.array("solutionVector").tobytes()).hexdigest()[:12])] += 1
print(n, dict(hist), flush=True)
