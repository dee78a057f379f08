#!/usr/bin/env python3
"""A/B of the tile apply (r05): k_tile_apply<0, 64> (PS_TILE_W=0) against the pipelined one-wave-per-tile form k_tile_apply_w (default from
~4096 tiles; PS_TILE_W=2 forces it).  Children print the SHA-1 of A x for a seeded x, the PCG iteration count and the kernel's time in
sequence; the parent compares.   usage: tile_ab.py [res]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if os.environ.get("TILE_AB_CHILD"):
    import numpy as np
    import polystokes_amd
    from polystokes_amd import scenes
    sc, p = scenes.cavity(res, tile=16, pad=2)
    s = polystokes_amd.Solver(0)
    s.upload(sc, p)
    rc = s.step_device()
    x = np.random.RandomState(3).standard_normal(s.nP + s.nT)
    y = s.apply(x)
    out = {"sha_Ax": hashlib.sha1(y.tobytes()).hexdigest(), "iters": int(s.stats.solveData[1]), "rc": int(rc), "sha_x": hashlib.sha1(s.array("solutionVector").tobytes()).hexdigest(),
           "tiles_ms_seq": [s.bench_kernel("seq:tiles", 20)[0] for _ in range(3)], "solve_ms": float(s.stats.stage_ms[8])}
    print("RESULT " + json.dumps(out), flush=True)
    s.close()
    sys.exit(0)
res_ = {}
for rnd in range(2):
    for w in ("0", "2"):
        env = dict(os.environ, PS_TILE_W=w, TILE_AB_CHILD="1")
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), str(res)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in pr.stdout.splitlines() if l.startswith("RESULT ")]
        assert line, pr.stdout[-2000:]
        d = json.loads(line[0][7:])
        res_.setdefault(w, []).append(d)
        print("round %d PS_TILE_W=%s: tiles %s ms, solve %.1f ms, %d iterations, A x %s, x %s" % (rnd, w, ["%.4f" % v for v in d["tiles_ms_seq"]], d["solve_ms"], d["iters"], d["sha_Ax"][:12], d["sha_x"][:12]), flush=True)
print("bit-identical A x:", res_["0"][0]["sha_Ax"] == res_["2"][0]["sha_Ax"], " bit-identical solution:", res_["0"][0]["sha_x"] == res_["2"][0]["sha_x"])
