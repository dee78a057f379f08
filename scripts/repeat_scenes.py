"""Determinism stress: repeated full steps (setup + solve) on several scenes, single rank and in-process groups; every
scene must yield exactly one (result, iterations, error, velocity checksum) tuple."""
import sys, os, collections, hashlib; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cases = {"blob": scenes.blob(seed=3), "blob_t7": scenes.blob(30, 26, 22, seed=1, tile=7, pad=2), "spheres40": scenes.spheres(40, tile=8),
         "coil48": scenes.coil(48), "droplet24": scenes.droplet(24), "beam32": scenes.beam(32), "cavity20_t10": scenes.cavity(20, tile=10, pad=1)}
bad = 0
s = polystokes_amd.Solver(0)
for name, (sc, p) in cases.items():
    for pre in (abi.PRE_IDENTITY, abi.PRE_DIAGONAL):
        p.preconditioner = pre
        s.upload(sc, p)
        seen = collections.Counter()
        for _ in range(reps):
            rc = s.step_device()
            vel, valid = s.download()
            h = hashlib.sha1(b"".join(v.tobytes() for v in vel) + b"".join(v.tobytes() for v in valid)).hexdigest()[:12]
            seen[(rc, int(s.stats.solveData[1]), float(s.stats.solveData[0]).hex(), h)] += 1
        print("OK " if len(seen) == 1 else "BAD", name, "pre", pre, dict(seen) if len(seen) > 1 else list(seen)[0][:2], flush=True)
        bad += len(seen) != 1
# in-process groups
sc0, p = scenes.cavity(32, precond=abi.PRE_DIAGONAL)
velx = np.zeros((64, 32, 33), np.float32); velx[63] = 1.0; velx[32, :, :16] = -0.5
tall = abi.Scene(32, 32, 64, sc0.dx, sc0.dt, 1.0, [velx, 0.0, 0.0], -1.0, 1.0, 1.0)
for world in (2, 4):
    seen = collections.Counter()
    for _ in range(max(reps // 3, 10)):
        g = polystokes_amd.Group(world)
        rc = g.solve_scene(tall, p)
        h = hashlib.sha1(b"".join(v.tobytes() for v in g.vel)).hexdigest()[:12]
        seen[(rc, int(g.stats.solveData[1]), float(g.stats.solveData[0]).hex(), h)] += 1
        g.close()
    print("OK " if len(seen) == 1 else "BAD", "group", world, dict(seen) if len(seen) > 1 else list(seen)[0][:2], flush=True)
    bad += len(seen) != 1
print("bad", bad)
sys.exit(1 if bad else 0)
