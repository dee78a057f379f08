"""Randomised check of the slab decomposition: in-process ranks (Group) against the single-domain solve on random scenes."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi, partition
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
single = polystokes_amd.Solver(0)
only = os.environ.get("FUZZ_ONLY")
for case in ([int(only)] if only else range(n_cases)):
    rng = np.random.RandomState(seed0 + case)
    world = int(rng.choice([2, 2, 3, 4]))
    tile = int(rng.choice([8, 16, 16]))
    nz = 16 * int(rng.randint(2 * world, 3 * world + 2))
    nx, ny = (int(v) for v in rng.randint(16, 40, 2))
    sc, p = scenes.blob(nx, ny, nz, seed=seed0 + case, tile=tile, pad=int(rng.choice([1, 2])), variable_viscosity=bool(rng.randint(2)))
    p.preconditioner = int(rng.choice([abi.PRE_IDENTITY, abi.PRE_DIAGONAL, abi.PRE_CHEBYSHEV]))
    p.activeLiquidBoundaryLayerSize = int(rng.choice([1, 2, 3])); p.activeSolidBoundaryLayerSize = int(rng.choice([0, 1, 2]))
    p.tolerance = float(os.environ.get("FUZZ_TOL", "1e-6"))
    p.maxSolverIterations = int(os.environ.get("FUZZ_MAXIT", "20000"))
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(world)
    msgs = []
    try:
        rc2 = grp.solve_scene(sc, p)
    except Exception as e:
        msgs.append("exception %s" % str(e)[:80]); rc2 = None
    if rc2 is not None:
        if rc1 != rc2: msgs.append("rc %d vs %d" % (rc1, rc2))
        it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
        # 5 %: where the stop rule fires on a plateau of an ill-conditioned solve (tol 1e-6, ~1000 iterations) moves by 4 % with the order
        # of the dot products alone — single domain, seed 6356: 1000 / 1041 / 998 iterations with PS_CHUNK_PLAIN / PS_XCD=0 / PS_PIPE_GRID=0
        if abs(it1 - it2) > max(3, 0.05 * it1): msgs.append("iters %d vs %d" % (it1, it2))
        lab = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx)
        for r, sl in enumerate(grp.slabs):
            ll = grp.ranks[r].array("centerLabels").reshape(sl.nz_local, sc.ny, sc.nx)
            if not np.array_equal(ll[sl.zLoOwned:sl.zHiOwned], lab[sl.z0:sl.z1]): msgs.append("labels rank %d" % r)
        for a in range(3):
            if not np.array_equal(grp.valid[a], single.valid[a]): msgs.append("valid%s" % "XYZ"[a])
            scale = max(np.abs(single.vel[a]).max(), 1e-30)
            dv = np.abs(grp.vel[a] - single.vel[a]).max() / scale
            if dv > 1000 * p.tolerance: msgs.append("vel%s %.1e" % ("XYZ"[a], dv))   # 1e-3 at the default tolerance of the sweep (AMP: DESIGN section 4)
    # LSQ (scripts/fuzz_parity.py): the per-tile fit systems are rank deficient; a slab computes a tile's offsets in its
    # own coordinates (the slab's z origin), the sums differ in the last bits and a borderline pivot of the rank-revealing LU can flip:
    # another — equally valid — fit vector, hence another right-hand side.  Detected by comparing the fit vectors of the owned tiles.
    lsq = False
    if rc2 is not None and msgs and all(m.startswith("vel") for m in msgs) and int(single.stats.dimData[24]) > 0:
        Rg = int(single.stats.dimData[24])
        cs = single.array("reducedRegionBestFitVectors").reshape(Rg, -1)
        rs = single.array("centerReducedIndices").reshape(sc.nz, sc.ny, sc.nx)
        for r, sl in enumerate(grp.slabs):
            rr = grp.ranks[r].array("centerReducedIndices").reshape(sl.nz_local, sc.ny, sc.nx)
            cl = grp.ranks[r].array("reducedRegionBestFitVectors")
            if cl.size == 0: continue
            cl = cl.reshape(-1, cs.shape[1])
            own, ref = rr[sl.zLoOwned:sl.zHiOwned], rs[sl.z0:sl.z1]
            for lq, gq in sorted(set(zip(own[own >= 0].tolist(), ref[own >= 0].tolist()))):
                if np.linalg.norm(cl[lq] - cs[gq]) > 1e-8 * max(np.linalg.norm(cs[gq]), 1e-300): lsq = True
    # AMP: two converged iterates of an ill-conditioned system (thousands of iterations at this tolerance) agree in x
    # to a multiple of the tolerance but their velocities — differences of large terms — to far less; single-GPU runs with another
    # summation order show the same spread (PS_CHUNK_PLAIN=1 on the same seeds).  Tagged, not counted.
    amp = bool(msgs) and all(m.startswith("vel") for m in msgs) and single.stats.solveData[1] > 500
    bad += bool(msgs) and not amp and not lsq
    print(("LSQ" if lsq else ("AMP" if amp else "BAD")) if msgs else "OK ", case, "world", world, (nx, ny, nz), "tile", tile, p.tilePadding, "pre", p.preconditioner,
          "dofs", int(single.stats.dimData[21]), "regions", int(single.stats.dimData[24]), "iters", int(single.stats.solveData[1]),
          int(grp.stats.solveData[1]) if rc2 is not None else -1, msgs, flush=True)
    grp.close()
print("cases", n_cases, "bad", bad)
sys.exit(1 if bad else 0)
