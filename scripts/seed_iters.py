import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import polystokes_amd
from helpers import fuzz_brick_case
from oracle import ps_oracle
seed = int(sys.argv[1]); tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
sc, p, dims, n, tile = fuzz_brick_case(seed, tol); p.maxSolverIterations = 100000
s = polystokes_amd.Solver(0); rc = s.step(sc, p)
print("env", {k: v for k, v in os.environ.items() if k.startswith("PS_")}, "rc", rc, "iters", int(s.stats.solveData[1]), "err %.3e" % s.stats.solveData[0], "fused", int(s.array("fusedStep")[0]), flush=True)
if len(sys.argv) > 3:
    o = ps_oracle.Oracle(); ro = o.run(sc, p); print("oracle rc", ro, "iters", int(o.stats.solveData[1]), "err %.3e" % o.stats.solveData[0])
