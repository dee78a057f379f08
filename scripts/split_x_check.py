"""PS_SPLIT_X against the four-kernel step: same iteration count, bit-identical solution vector?  usage: split_x_check.py [n]  (spawns itself per setting)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 2:
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    n = int(sys.argv[1])
    sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    s = polystokes_amd.Solver(0)
    rc = s.step(sc, p)
    np.save(sys.argv[2], s.array("solutionVector"))
    print(os.environ.get("PS_SPLIT_X", "0"), "rc", rc, "iters", int(s.stats.solveData[1]), "err", s.stats.solveData[0], "solve ms", s.stats.solveData[3], "fused", int(s.array("fusedStep")[0]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "128"
outs = []
for v in ("0", "1", "2"):
    f = "/tmp/split_%s.npy" % v
    subprocess.check_call([sys.executable, os.path.abspath(__file__), n, f], env=dict(os.environ, PS_SPLIT_X=v))
    outs.append(np.load(f))
for v, o in zip(("1", "2"), outs[1:]):
    print("PS_SPLIT_X=%s vs 0: identical" % v if np.array_equal(o, outs[0]) else "PS_SPLIT_X=%s vs 0: max diff %.3e" % (v, np.abs(o - outs[0]).max()))
