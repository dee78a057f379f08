"""Randomised check of the brick decomposition (ps_set_brick): in-process ranks against the single-domain solve on random blob scenes.
usage: fuzz_bricks.py [cases] [seed0]     (FUZZ_TOL, FUZZ_ONLY as in fuzz_multirank.py)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = 0
single = polystokes_amd.Solver(0)
only = os.environ.get("FUZZ_ONLY")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import fuzz_brick_case          # the case generator lives with the tests (tests/test_gpu_multirank.py replays the hard seeds)
for case in ([int(only)] if only else range(n_cases)):
    sc, p, dims, n, tile = fuzz_brick_case(seed0 + case, float(os.environ.get("FUZZ_TOL", "1e-6")))
    rc1 = single.step(sc, p)
    world = dims[0] * dims[1] * dims[2]
    grp = polystokes_amd.Group(world, dims=dims)
    msgs = []
    try:
        rc2 = grp.solve_scene(sc, p)
    except Exception as e:
        msgs.append("exception %s" % str(e)[:100]); rc2 = None
    if rc2 is not None:
        if rc1 != rc2: msgs.append("rc %d vs %d" % (rc1, rc2))
        it1, it2 = single.stats.solveData[1], grp.stats.solveData[1]
        if abs(it1 - it2) > max(3, 0.05 * it1): msgs.append("iters %d vs %d" % (it1, it2))
        lab = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx)
        for r, b in enumerate(grp.bricks):
            ll = grp.ranks[r].array("centerLabels").reshape(b.n_local[2], b.n_local[1], b.n_local[0])
            if not np.array_equal(ll[b.lo[2]:b.hi[2], b.lo[1]:b.hi[1], b.lo[0]:b.hi[0]], lab[b.g0[2]:b.g1[2], b.g0[1]:b.g1[1], b.g0[0]:b.g1[0]]): msgs.append("labels rank %d" % r)
        for a in range(3):
            if not np.array_equal(grp.valid[a], single.valid[a]): msgs.append("valid%s" % "XYZ"[a])
            scale = max(np.abs(single.vel[a]).max(), 1e-30)
            dv = np.abs(grp.vel[a] - single.vel[a]).max() / scale
            if dv > 1000 * p.tolerance: msgs.append("vel%s %.1e" % ("XYZ"[a], dv))
    grp.close()
    tag = "ok" if not msgs else "MISMATCH " + "; ".join(msgs)
    if msgs: bad += 1
    print("case %d seed %d grid %s dims %s tile %d pre %d L%d S%d iters %s: %s" % (case, seed0 + case, n, dims, tile, p.preconditioner, p.activeLiquidBoundaryLayerSize,
          p.activeSolidBoundaryLayerSize, int(single.stats.solveData[1]), tag), flush=True)
print("mismatches: %d of %d" % (bad, n_cases))
