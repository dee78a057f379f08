"""The least-squares fit of the tiles (reducedRegionBestFitVectors, reducedRHSVector) of two libraries bit for bit + tile-matrices stage time.
usage: lsq_ab.py <libA.so> <libB.so> [scene res ...]   (spawns itself per library)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
NAMES = ["reducedRegionBestFitVectors", "reducedRHSVector", "reducedMassMatrices", "reducedViscosityMatrices"]
if sys.argv[1] == "--child":
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    scene, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    sc, p = getattr(scenes, scene)(n)
    s = polystokes_amd.Solver(0); s.upload(sc, p)
    best = 1e9
    for _ in range(3):
        s.setup(); best = min(best, float(s.stats.stage_ms[4]))
    np.savez(out, **{k: s.array(k) for k in NAMES})
    print(os.environ.get("PS_LIB", "(default)"), scene, n, "tile matrices stage ms %.3f" % best, flush=True)
    sys.exit(0)
libs = sys.argv[1:3]
cases = sys.argv[3:] or ["cavity", "128"]
for k in range(0, len(cases), 2):
    outs = []
    for i, lib in enumerate(libs):
        f = "/tmp/lsq_%d.npz" % i
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", cases[k], cases[k + 1], f], env=dict(os.environ, PS_LIB=lib))
        outs.append(np.load(f))
    for n in NAMES:
        a, b = outs[0][n], outs[1][n]
        print(" ", cases[k], cases[k + 1], n, "identical" if np.array_equal(a, b) else "max rel diff %.3e" % (np.abs(a - b).max() / max(np.abs(a).max(), 1e-300)), flush=True)
