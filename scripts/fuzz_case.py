import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
from oracle import ps_oracle
seed0, case = int(sys.argv[1]), int(sys.argv[2])
os.environ.setdefault("FUZZ_MAX", "72")
rng = np.random.RandomState(seed0 + case)
nx, ny, nz = (int(v) for v in rng.randint(14, int(os.environ["FUZZ_MAX"]), 3))
tile = int(rng.choice([5, 6, 7, 8, 9, 10, 12, 16])); pad = int(rng.choice([1, 2, 2, 3]))
sc, p = scenes.blob(nx, ny, nz, seed=seed0 + case, tile=tile, pad=min(pad, tile - 1), variable_viscosity=bool(rng.randint(2)))
p.activeLiquidBoundaryLayerSize = int(rng.choice([1, 2, 2, 3])); p.activeSolidBoundaryLayerSize = int(rng.choice([0, 1, 2, 2, 3]))
p.doTile = int(rng.rand() < 0.85); p.doReducedRegions = int(rng.rand() < 0.9)
p.indexOrder = int(rng.choice([abi.ORDER_VOXEL_TILES, abi.ORDER_VOXEL_TILES, abi.ORDER_LINEAR])); p.preconditioner = int(rng.choice([abi.PRE_IDENTITY, abi.PRE_DIAGONAL]))
if len(sys.argv) > 3: p.tolerance = float(sys.argv[3])
o = ps_oracle.Oracle(); o.run(sc, p, solve=True)
g = polystokes_amd.Solver(0); rc = g.step(sc, p)
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
print("dims", (nx, ny, nz), "tol", p.tolerance, "iters", g.stats.solveData[1], o.stats.solveData[1], "err", g.stats.solveData[0], o.stats.solveData[0])
for nm in ("b", "solutionVector", "activeRHSVector", "reducedRHSVector", "Inv_Mr_plus_2JDtuDJ", "reducedMassMatrices", "reducedViscosityMatrices", "reducedRegionBestFitVectors",
           "recoveredActiveVelocity", "recoveredReducedVelocity"):
    try:
        print("  %-30s rel diff %.3e   |ref| %.3e" % (nm, rel(g.array(nm), o.array(nm)), np.linalg.norm(o.array(nm))))
    except Exception as e:
        print("  ", nm, "n/a", e)
for a in range(3):
    ref = o.array("vel" + "XYZ"[a]).reshape(g.vel[a].shape)
    d = np.abs(g.vel[a] - ref)
    i = np.unravel_index(d.argmax(), d.shape)
    red = g.array("face%sReducedIndices" % "XYZ"[a]).reshape(d.shape)
    lab = g.array("face%sLabels" % "XYZ"[a]).reshape(d.shape)
    print("  vel%s max|diff| %.3e at %s (gpu %.5g ref %.5g) scale %.3e  label %d region %d ; max diff on reduced faces %.3e, on others %.3e" % (
        "XYZ"[a], d.max(), i, g.vel[a][i], ref[i], np.abs(ref).max(), lab[i], red[i], d[lab == abi.REDUCED].max() if (lab == abi.REDUCED).any() else 0, d[lab != abi.REDUCED].max()))
R = o.nRegions
if R:
    N = o.array("reducedRegionBestFitSystems").reshape(R, 26, 26); rhsN = o.array("reducedRegionBestFitRHS").reshape(R, 26)
    cg_, co_ = g.array("reducedRegionBestFitVectors").reshape(R, 26), o.array("reducedRegionBestFitVectors").reshape(R, 26)
    for r in range(R):
        sv = np.linalg.svd(N[r], compute_uv=False)
        d = np.linalg.norm(cg_[r] - co_[r]) / max(np.linalg.norm(co_[r]), 1e-300)
        if d > 1e-8 or r < 3:
            ls = np.linalg.lstsq(N[r], rhsN[r], rcond=1e-12)[0]
            print("  region %3d  sv max %.3e min %.3e  rank(1e-10) %d  |cfit gpu-oracle|/|oracle| %.2e  |oracle| %.3e  |min-norm lsq| %.3e  resid gpu %.2e oracle %.2e" % (
                r, sv[0], sv[-1], int((sv > 1e-10 * sv[0]).sum()), d, np.linalg.norm(co_[r]), np.linalg.norm(ls),
                np.linalg.norm(N[r] @ cg_[r] - rhsN[r]) / max(np.linalg.norm(rhsN[r]), 1e-300), np.linalg.norm(N[r] @ co_[r] - rhsN[r]) / max(np.linalg.norm(rhsN[r]), 1e-300)))
