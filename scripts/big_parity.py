"""One-off parity check at a larger size than the test suite affords: GPU against the CPU oracle (minutes of CPU)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
from oracle import ps_oracle
for spec in sys.argv[1:]:
    name, n = spec.rstrip("0123456789"), int(spec[len(spec.rstrip("0123456789")):])
    sc, p = getattr(scenes, name)(n)
    t0 = time.time(); o = ps_oracle.Oracle(); o.run(sc, p, solve=True); to = time.time() - t0
    g = polystokes_amd.Solver(0); t0 = time.time(); rc = g.step(sc, p); tg = time.time() - t0
    bad = []
    if list(g.stats.dimData) != list(o.stats.dimData): bad.append("dimData")
    for s in abi.SAMPLE_NAMES:
        for kind in ("LiquidWeights", "FluidWeights", "Labels", "ActiveIndices", "ReducedIndices"):
            if not np.array_equal(g.array(s + kind), o.array(s + kind)): bad.append(s + kind)
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
    v = np.random.RandomState(0).standard_normal(o.nP + o.nT)
    print(spec, "oracle %.0fs gpu %.2fs" % (to, tg), "rc", rc, o.result, "iters", int(g.stats.solveData[1]), int(o.stats.solveData[1]),
          "dofs", o.nP + o.nT, "regions", o.nRegions, "int-state", "bit-exact" if not bad else bad,
          "| b %.1e apply %.1e x %.1e cfit %.1e" % (rel(g.array("b"), o.array("b")), rel(g.apply(v), o.apply(v)), rel(g.array("solutionVector"), o.array("solutionVector")),
                                                   rel(g.array("reducedRegionBestFitVectors"), o.array("reducedRegionBestFitVectors")) if o.nRegions else 0.0),
          "vel max diff", ["%.1e" % (np.abs(g.vel[a] - o.array("vel" + "XYZ"[a]).reshape(g.vel[a].shape)).max()) for a in range(3)], flush=True)
    g.close()
