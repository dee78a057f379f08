"""Two libraries (PS_LIB) on the same scenes: registered setup arrays and the solution bit for bit, stage times side by side.
usage: setup_ab.py <libA.so> <libB.so> [scene res ...]   (spawns itself per library)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
NAMES = ["S.chunkInfo", "S.chunkRep", "S.code", "St.chunkInfo", "St.chunkRep", "St.code", "streamRuns", "centerReducedIndices", "dinv", "solutionVector"]
STAGES = ["weights", "classify", "regions", "indices", "tile_matrices", "blocks", "assemble", "precond"]
if sys.argv[1] == "--child":
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    scene, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    sc, p = getattr(scenes, scene)(n)
    p.preconditioner = abi.PRE_DIAGONAL
    s = polystokes_amd.Solver(0); s.upload(sc, p)
    best = None
    for _ in range(3):
        s.setup()
        st = [float(s.stats.stage_ms[i]) for i in range(8)]
        best = st if best is None else [min(a, b) for a, b in zip(best, st)]
    rc = s.step(sc, p)
    np.savez(out, **{k: s.array(k) for k in NAMES})
    print(os.environ.get("PS_LIB", "(default)"), scene, n, " ".join("%s %.2f" % (a, b) for a, b in zip(STAGES, best)), "| setup %.2f" % sum(best), "| iters", int(s.stats.solveData[1]), flush=True)
    sys.exit(0)
libs = sys.argv[1:3]
cases = sys.argv[3:] or ["cavity", "128"]
for k in range(0, len(cases), 2):
    outs = []
    for i, lib in enumerate(libs):
        f = "/tmp/sab_%d.npz" % i
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", cases[k], cases[k + 1], f], env=dict(os.environ, PS_LIB=lib))
        outs.append(np.load(f))
    bad = [n for n in NAMES if not np.array_equal(outs[0][n], outs[1][n])]
    print(" ", cases[k], cases[k + 1], "all %d arrays identical" % len(NAMES) if not bad else "DIFFER: %s" % bad, flush=True)
