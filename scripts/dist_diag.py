"""Where a slab group's velocities differ from the single-domain solve (fuzz_multirank scene by seed, world 3)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
seed = int(sys.argv[1]); w = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.RandomState(seed)
world = int(rng.choice([2, 2, 3, 4])); tile = int(rng.choice([8, 16, 16])); nz = 16 * int(rng.randint(2 * world, 3 * world + 2))
nx, ny = (int(v) for v in rng.randint(16, 40, 2))
pad = int(rng.choice([1, 2])); vv = bool(rng.randint(2))
sc, p = scenes.blob(nx, ny, nz, seed=seed, tile=tile, pad=pad, variable_viscosity=vv)
p.preconditioner = int(rng.choice([abi.PRE_IDENTITY, abi.PRE_DIAGONAL, abi.PRE_CHEBYSHEV]))
p.activeLiquidBoundaryLayerSize = int(rng.choice([1, 2, 3])); p.activeSolidBoundaryLayerSize = int(rng.choice([0, 1, 2]))
p.tolerance = 1e-6; p.maxSolverIterations = 20000; p.preconditioner = abi.PRE_DIAGONAL
print("grid", nx, ny, nz, "tile", tile, "pad", pad, "L/S", p.activeLiquidBoundaryLayerSize, p.activeSolidBoundaryLayerSize, "varvisc", vv)
single = polystokes_amd.Solver(0); single.step(sc, p)
grp = polystokes_amd.Group(w); grp.solve_scene(sc, p)
print("slabs", [(s.z0, s.z1) for s in grp.slabs], "regions single", int(single.stats.dimData[24]))
for a in range(3):
    d = np.abs(grp.vel[a] - single.vel[a])
    idx = np.unravel_index(np.argmax(d), d.shape)
    lab = single.array("face%sLabels" % "XYZ"[a]).reshape(d.shape)
    red = single.array("face%sReducedIndices" % "XYZ"[a]).reshape(d.shape)
    big = np.argwhere(d > 1e-3 * np.abs(single.vel[a]).max())
    print("axis", a, "max diff", d.max(), "at (z,y,x)", idx, "label", lab[idx], "region", red[idx], "count > 1e-3:", len(big),
          "z range", (big[:, 0].min(), big[:, 0].max()) if len(big) else None, "regions involved", sorted(set(red[tuple(b)] for b in big))[:8])
print("single n", single.nP + single.nT, "iters", single.stats.solveData[1], "group iters", grp.stats.solveData[1])
xs = single.array("solutionVector"); bs = single.array("b")
for r, s in enumerate(grp.ranks):
    n = s.nP + s.nT
    print("rank", r, "local DOFs", n, "regions", s.nRegions, "owned range?", end=" ")
    if n == xs.size:
        xr = s.array("solutionVector"); br = s.array("b")
        print("x rel diff %.2e  b rel diff %.2e" % (np.linalg.norm(xr - xs) / np.linalg.norm(xs), np.linalg.norm(br - bs) / np.linalg.norm(bs)))
    else:
        print()
