"""Distributed-algorithm overhead without a network: `world` in-process ranks on ONE GPU (they run one after the other on one
stream), each with an n^2 x n slab of the weak-scaling cavity, against one rank with the n^3 cavity."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
worlds = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2]
sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0); s.upload(sc, p); s.step_device()
t0 = time.perf_counter(); s.step_device(); t1 = time.perf_counter() - t0
it1 = int(s.stats.solveData[1]); solve1 = s.stats.stage_ms[8]
print("single", n, "iters", it1, "step %.1f ms" % (t1 * 1e3), "solve %.1f ms" % solve1, "per-iter %.3f ms" % (solve1 / max(it1 + 1, 1)), flush=True)
s.close()
for world in worlds:
    g = polystokes_amd.Group(world)
    for r in range(world):
        scr, pr, sl = scenes.cavity_slab(n, world, r, precond=abi.PRE_DIAGONAL)
        g.ranks[r].upload(scr, pr); g.ranks[r].set_slab(sl)
    g.step()
    t0 = time.perf_counter(); rc = g.step(); tw = time.perf_counter() - t0
    itw = int(g.stats.solveData[1]); solvew = g.stats.solveData[3]
    print("group", world, "rc", rc, "iters", itw, "step %.1f ms" % (tw * 1e3), "solve %.1f ms" % solvew,
          "per-iter per-rank %.3f ms" % (solvew / max(itw + 1, 1) / world), "overhead vs single per-iter x%.3f" % ((solvew / max(itw + 1, 1) / world) / (solve1 / max(it1 + 1, 1))), flush=True)
    g.close()
