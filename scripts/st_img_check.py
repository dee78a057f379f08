"""EXPERIMENT PS_ST_IMG (groups of N chunks gathered from an LDS image): A x bit for bit against the default kernels, and the isolated
St launch time.  usage: st_img_check.py [n] [gc ...]   (spawns itself per setting)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    n, out = int(sys.argv[2]), sys.argv[3]
    sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    s = polystokes_amd.Solver(0); s.upload(sc, p); s.setup()
    x = np.random.RandomState(3).standard_normal(s.nP + s.nT)
    y = s.apply(x)
    np.save(out, y)
    ms = min(s.bench_kernel("spmv_St", 30)[0] for _ in range(3))
    ms_seq = s.bench_kernel("seq:spmv_St", 20)[0]
    print("PS_ST_IMG=%s" % os.environ.get("PS_ST_IMG", "0"), "n", n, "spmv_St isolated %.4f ms  in sequence %.4f ms" % (ms, ms_seq), "setup ms %.1f" % float(s.stats.solveData[5]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "128"
gcs = sys.argv[2:] or ["4", "8"]
outs = []
for v in ["0"] + gcs:
    f = "/tmp/stimg_%s.npy" % v
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", n, f], env=dict(os.environ, PS_ST_IMG=v, PS_VERBOSE="1"))
    outs.append(np.load(f))
for v, o in zip(gcs, outs[1:]):
    print("PS_ST_IMG=%s vs default: %s" % (v, "identical" if np.array_equal(o, outs[0]) else "max rel diff %.3e" % (np.abs(o - outs[0]).max() / np.abs(outs[0]).max())))
