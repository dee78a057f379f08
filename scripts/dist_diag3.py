"""Region partition of the owned cells: every rank of a slab group against the single domain (fuzz_multirank scene by seed)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
seed = int(sys.argv[1]); w = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.RandomState(seed)
world = int(rng.choice([2, 2, 3, 4])); tile = int(rng.choice([8, 16, 16])); nz = 16 * int(rng.randint(2 * world, 3 * world + 2))
nx, ny = (int(v) for v in rng.randint(16, 40, 2))
pad = int(rng.choice([1, 2])); vv = bool(rng.randint(2))
rng.choice([1, 5, 6]); L = int(rng.choice([1, 2, 3])); S = int(rng.choice([0, 1, 2]))
sc, p = scenes.blob(nx, ny, nz, seed=seed, tile=tile, pad=pad, variable_viscosity=vv)
p.preconditioner = abi.PRE_DIAGONAL; p.tolerance = 1e-6; p.maxSolverIterations = 20000
p.activeLiquidBoundaryLayerSize, p.activeSolidBoundaryLayerSize = L, S
single = polystokes_amd.Solver(0); single.step(sc, p)
grp = polystokes_amd.Group(w); grp.solve_scene(sc, p)
rs = single.array("centerReducedIndices").reshape(nz, ny, nx)
ls = single.array("centerLabels").reshape(nz, ny, nx)
coms = single.array("reducedRegionCOM").reshape(-1, 3)
print("single regions", coms.shape[0], "sizes", [int((rs == r).sum()) for r in range(coms.shape[0])])
for r, sl in enumerate(grp.slabs):
    s = grp.ranks[r]
    rr = s.array("centerReducedIndices").reshape(sl.nz_local, ny, nx)
    com = s.array("reducedRegionCOM").reshape(-1, 3)
    own = rr[sl.zLoOwned:sl.zHiOwned]; ref = rs[sl.z0:sl.z1]
    print("rank", r, "z", (sl.z0, sl.z1), "local regions", com.shape[0], "sizes (whole local domain)", [int((rr == q).sum()) for q in range(com.shape[0])])
    pairs = sorted(set(zip(own[own >= 0].tolist(), ref[own >= 0].tolist())))
    print("   (local region, global region) pairs on owned cells:", pairs, " mismatched membership:", int(((own >= 0) != (ref >= 0)).sum()))
    for lq, gq in pairs:
        print("   region %d->%d  COM (a slab reports global coordinates) %s  single domain %s  cells local %d global %d" % (lq, gq, np.round(com[lq], 6), np.round(coms[gq], 6), int((rr == lq).sum()), int((rs == gq).sum())))
s1 = grp.ranks[1]
for nm in ("reducedMassMatrices", "reducedViscosityMatrices", "Inv_Mr_plus_2JDtuDJ", "reducedRHSVector"):
    try:
        a, b = single.array(nm), s1.array(nm)
        print(nm, a.shape, b.shape, "rel diff %.2e" % (np.abs(a - b).max() / max(np.abs(a).max(), 1e-300)) if a.shape == b.shape else "shape mismatch")
    except Exception as e:
        print(nm, "error", e)
