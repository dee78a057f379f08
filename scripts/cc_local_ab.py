"""PS_CC_LOCAL A/B: region ids, labels and DOF indices bit for bit, regions stage time.  usage: cc_local_ab.py [scene res ...]  (spawns itself)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
NAMES = ["centerLabels", "centerReducedIndices", "faceXReducedIndices", "faceYReducedIndices", "faceZReducedIndices"]
if sys.argv[1] == "--child":
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    scene, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    kw = {}
    if len(sys.argv) > 5: kw["tile"] = int(sys.argv[5])
    sc, p = getattr(scenes, scene)(n, **kw)
    s = polystokes_amd.Solver(0); s.upload(sc, p)
    best = 1e9
    for _ in range(3):
        s.setup(); best = min(best, float(s.stats.stage_ms[2]))
    np.savez(out, **{k: s.array(k) for k in NAMES})
    print("PS_CC_LOCAL=%s" % os.environ.get("PS_CC_LOCAL", "1"), scene, n, kw, "regions stage ms %.3f" % best, "regions", int(s.stats.dimData[24]), flush=True)
    sys.exit(0)
cases = sys.argv[1:] or ["cavity", "128"]
for k in range(0, len(cases), 2):
    outs = []
    for v in ("0", "1"):
        f = "/tmp/ccl_%s.npz" % v
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", cases[k].split(":")[0], cases[k + 1], f] + cases[k].split(":")[1:], env=dict(os.environ, PS_CC_LOCAL=v))
        outs.append(np.load(f))
    print(cases[k], cases[k + 1], "identical" if all(np.array_equal(outs[0][n], outs[1][n]) for n in NAMES) else "DIFFER", flush=True)
