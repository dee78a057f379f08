#!/usr/bin/env python3
"""In-process brick decompositions against the single domain: iterations and velocity differences.  usage: brick_case.py scene res dx dy dz [precond]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
scene, n = sys.argv[1], int(sys.argv[2])
dims = tuple(int(v) for v in sys.argv[3:6])
sc, p = getattr(scenes, scene)(n, tile=16)
if len(sys.argv) > 6:
    p.preconditioner = int(sys.argv[6])
single = polystokes_amd.Solver(0)
single.step(sc, p)
grp = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
rc = grp.solve_scene(sc, p)
e = [float(np.abs(grp.vel[a] - single.vel[a]).max() / max(np.abs(single.vel[a]).max(), 1e-30)) for a in range(3)]
print("solve ms single %.1f group %.1f (in-process ranks run one after the other: x%.3f of the single domain)" % (single.stats.stage_ms[8], grp.stats.solveData[3], grp.stats.solveData[3] / max(single.stats.stage_ms[8], 1e-9)))
print(scene, n, dims, "rc", rc, "iterations", single.stats.solveData[1], grp.stats.solveData[1], "vel err", e, "valid equal", [bool(np.array_equal(grp.valid[a], single.valid[a])) for a in range(3)])
