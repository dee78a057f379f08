"""Velocity spread between converged solves of ONE fuzz_multirank scene (seed argument): single domain with identity / Jacobi / Chebyshev,
at the sweep's tolerance and at a 100x tighter one.  Separates AMP from a wrong distributed solve."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
seed = int(sys.argv[1])
rng = np.random.RandomState(seed)
world = int(rng.choice([2, 2, 3, 4])); tile = int(rng.choice([8, 16, 16])); nz = 16 * int(rng.randint(2 * world, 3 * world + 2))
nx, ny = (int(v) for v in rng.randint(16, 40, 2))
sc, p = scenes.blob(nx, ny, nz, seed=seed, tile=tile, pad=int(rng.choice([1, 2])), variable_viscosity=bool(rng.randint(2)))
p.preconditioner = int(rng.choice([abi.PRE_IDENTITY, abi.PRE_DIAGONAL, abi.PRE_CHEBYSHEV]))
p.activeLiquidBoundaryLayerSize = int(rng.choice([1, 2, 3])); p.activeSolidBoundaryLayerSize = int(rng.choice([0, 1, 2]))
p.maxSolverIterations = 200000
s = polystokes_amd.Solver(0)
ref = None
for tol in (1e-6, 1e-8, 1e-10):
    for pre in (abi.PRE_IDENTITY, abi.PRE_DIAGONAL, abi.PRE_CHEBYSHEV):
        p.tolerance, p.preconditioner = tol, pre
        rc = s.step(sc, p)
        v = [np.array(a, copy=True) for a in s.vel]
        if tol == 1e-10 and pre == abi.PRE_DIAGONAL: ref = v
        print("tol %.0e pre %d rc %d iters %d" % (tol, pre, rc, int(s.stats.solveData[1])), end="")
        if ref is None: keep = v if (tol == 1e-6 and pre == abi.PRE_IDENTITY) else keep
        print("  vel vs first: %s" % ["%.1e" % (np.abs(v[a] - keep[a]).max() / max(np.abs(keep[a]).max(), 1e-30)) for a in range(3)], flush=True)
