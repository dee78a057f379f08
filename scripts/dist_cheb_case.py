"""One fuzz_multirank scene (seed): in-process slab groups against the single-domain solve for several preconditioners / degrees."""
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
p.tolerance = 1e-6; p.maxSolverIterations = 20000
single = polystokes_amd.Solver(0)
for pre, deg in ((abi.PRE_DIAGONAL, 0), (abi.PRE_CHEBYSHEV, 1), (abi.PRE_CHEBYSHEV, 2), (abi.PRE_CHEBYSHEV, 4)):
    p.preconditioner, p.preconditionerDegree = pre, deg
    rc1 = single.step(sc, p)
    for w in (2, 3):
        if nz // 16 < w: continue
        grp = polystokes_amd.Group(w)
        rc2 = grp.solve_scene(sc, p)
        dv = [np.abs(grp.vel[a] - single.vel[a]).max() / max(np.abs(single.vel[a]).max(), 1e-30) for a in range(3)]
        print("pre %d deg %d world %d rc %d %d iters %d %d vel %s" % (pre, deg, w, rc1, rc2, single.stats.solveData[1], grp.stats.solveData[1], ["%.1e" % d for d in dv]), flush=True)
        grp.close()
