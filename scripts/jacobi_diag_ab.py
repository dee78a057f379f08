"""k_jacobi_diag A/B: the Jacobi diagonal of two libraries (PS_LIB) bit for bit, and the preconditioner stage time.
usage: jacobi_diag_ab.py <libA.so> <libB.so> [scene res ...]   (spawns itself per library)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "--child":
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    scene, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    sc, p = getattr(scenes, scene)(n)
    p.preconditioner = abi.PRE_DIAGONAL
    s = polystokes_amd.Solver(0); s.upload(sc, p)
    best = 1e9
    for _ in range(3):
        s.setup(); best = min(best, float(s.stats.stage_ms[7]))
    np.save(out, s.array("dinv"))
    print(os.environ.get("PS_LIB", "(default)"), scene, n, "precond stage ms %.3f" % best, flush=True)
    sys.exit(0)
libs = sys.argv[1:3]
cases = sys.argv[3:] or ["cavity", "128"]
for k in range(0, len(cases), 2):
    outs = []
    for i, lib in enumerate(libs):
        f = "/tmp/jd_%d.npy" % i
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", cases[k], cases[k + 1], f], env=dict(os.environ, PS_LIB=lib))
        outs.append(np.load(f))
    print(cases[k], cases[k + 1], "identical" if np.array_equal(outs[0], outs[1]) else "DIFFER max %.3e" % np.abs(outs[0] - outs[1]).max(), flush=True)
