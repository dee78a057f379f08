#!/usr/bin/env python3
"""Same-box A/B of one environment switch of the lab build: children alternate over the values, two rounds; each prints the step time of the
256^3 cavity (3 warm steps), the solve stage, iterations, the in-sequence kernel times and the SHA-1 of the solution.
usage: env_ab.py VAR v0 v1 [res] [precond]"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
var, vals = sys.argv[1], sys.argv[2:4]
res = int(sys.argv[4]) if len(sys.argv) > 4 else 256
pre = sys.argv[5] if len(sys.argv) > 5 else "jacobi"
if os.environ.get("ENV_AB_CHILD"):
    import numpy as np
    import polystokes_amd
    from polystokes_amd import scenes, _abi as abi
    sc, p = scenes.cavity(res, tile=16, pad=2, precond={"jacobi": abi.PRE_DIAGONAL, "identity": abi.PRE_IDENTITY, "chebyshev": abi.PRE_CHEBYSHEV, "chebyshev32": abi.PRE_CHEBYSHEV_F32}[pre])
    s = polystokes_amd.Solver(0)
    s.upload(sc, p)
    s.step_device()
    t0 = time.time()
    for _ in range(3):
        rc = s.step_device()
    ms = (time.time() - t0) * 1e3 / 3
    out = {"ms_per_step": ms, "solve_ms": float(s.stats.stage_ms[8]), "iters": int(s.stats.solveData[1]), "rc": int(rc),
           "sha_x": hashlib.sha1(s.array("solutionVector").tobytes()).hexdigest()[:12],
           "kernels_us": {k: round(s.bench_kernel("seq:" + k, 20)[0] * 1e3, 1) for k in ("spmv_S", "tiles", "spmv_St_r", "cg_update_xp_u")}}
    print("RESULT " + json.dumps(out), flush=True)
    s.close()
    sys.exit(0)
for rnd in range(2):
    for v in vals:
        env = dict(os.environ, ENV_AB_CHILD="1")
        env[var] = v
        pr = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in pr.stdout.splitlines() if l.startswith("RESULT ")]
        assert line, pr.stdout[-3000:]
        d = json.loads(line[0][7:])
        print("round %d %s=%s: %.1f ms/step, solve %.1f ms, %d iterations (%.4f ms each), x %s, kernels %s" % (rnd, var, v, d["ms_per_step"], d["solve_ms"], d["iters"], d["solve_ms"] / max(d["iters"], 1), d["sha_x"], d["kernels_us"]), flush=True)
