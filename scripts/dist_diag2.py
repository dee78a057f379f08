"""Variants of one fuzz_multirank scene (seed) in a 3-rank group against the single domain: which ingredient makes them differ."""
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
single = polystokes_amd.Solver(0)
def run(tag, vvisc, red, l, s_, tl, pd, zero_cvel=False, zero_vel=False):
    sc, p = scenes.blob(nx, ny, nz, seed=seed, tile=tl, pad=pd, variable_viscosity=vvisc)
    p.preconditioner = abi.PRE_DIAGONAL; p.tolerance = 1e-6; p.maxSolverIterations = 20000
    p.activeLiquidBoundaryLayerSize, p.activeSolidBoundaryLayerSize, p.doReducedRegions = l, s_, red
    if zero_cvel and sc.collisionvel is not None: sc.collisionvel = [np.zeros_like(a) for a in sc.collisionvel]
    rc1 = single.step(sc, p)
    grp = polystokes_amd.Group(w); rc2 = grp.solve_scene(sc, p)
    dv = max(np.abs(grp.vel[a] - single.vel[a]).max() / max(np.abs(single.vel[a]).max(), 1e-30) for a in range(3))
    print("%-28s rc %d %d iters %d %d regions %d  max vel diff %.1e" % (tag, rc1, rc2, single.stats.solveData[1], grp.stats.solveData[1], single.stats.dimData[24], dv), flush=True)
    grp.close()
run("as drawn", vv, 1, L, S, tile, pad)
run("constant viscosity", False, 1, L, S, tile, pad)
run("no reduced regions", vv, 0, L, S, tile, pad)
run("L=2 S=2", vv, 1, 2, 2, tile, pad)
run("L=1 S=0", vv, 1, 1, 0, tile, pad)
run("pad 1", vv, 1, L, S, tile, 1)
run("tile 8", vv, 1, L, S, 8, pad)
run("no collision velocity", vv, 1, L, S, tile, pad, zero_cvel=True)
