#!/usr/bin/env python3
"""In-process PROXY for the cost of a decomposition (not a scaling curve: all ranks share one GPU and one stream): ONE res^3 scene solved by a
single domain and cut 8 ways — 8 z-slabs, 2 x 2 x 2 bricks — per rank and iteration against the single domain's iteration / 8.
usage: cut_overhead.py [scene] [res] [dims,dims,...]   e.g. cavity 256 1x1x8,2x2x2"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cuts = [tuple(int(v) for v in d.split("x")) for d in (sys.argv[3] if len(sys.argv) > 3 else "1x1x8,2x2x2").split(",")]
sc, p = getattr(scenes, scene)(res, tile=16, pad=2)
p.preconditioner = abi.PRE_DIAGONAL
s = polystokes_amd.Solver(0)
s.upload(sc, p); s.step_device(); s.step_device()
it1, solve1 = int(s.stats.solveData[1]), float(s.stats.stage_ms[8])
per1 = solve1 / max(it1 + 1, 1)
print("%s %d^3 single domain: %d iterations, solve %.1f ms, %.4f ms per iteration" % (scene, res, it1, solve1, per1), flush=True)
s.close()
for dims in cuts:
    w = dims[0] * dims[1] * dims[2]
    g = polystokes_amd.Group(w, dims=dims)
    g.solve_scene(sc, p)
    t0 = time.perf_counter(); rc = g.solve_scene(sc, p); el = (time.perf_counter() - t0) * 1e3
    it, solve = int(g.stats.solveData[1]), float(g.stats.solveData[3])
    per = solve / max(it + 1, 1) / w
    print("%dx%dx%d in-process ranks: rc %d, %d iterations, solve %.1f ms, %.4f ms per rank-iteration = x%.3f of (single domain / %d)" % (dims + (rc, it, solve, per, per / (per1 / w), w)), flush=True)
    g.close()
