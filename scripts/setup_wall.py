#!/usr/bin/env python3
"""Wall time of the setup stages (sum of stage_ms[0..7] + the host's clock around ps_setup_device) over a few setups.  usage: setup_wall.py scene res [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sc, p = getattr(scenes, name)(n); p.preconditioner = abi.PRE_DIAGONAL
s = polystokes_amd.Solver(0); s.upload(sc, p)
for _ in range(3): s.setup()
w = []
for _ in range(reps):
    t0 = time.perf_counter(); s.setup(); w.append((time.perf_counter() - t0) * 1e3)
w.sort()
print(os.environ.get("PS_LIB", "(default)"), name, n, "setup wall ms: min %.3f median %.3f" % (w[0], w[len(w) // 2]), "stages", [round(float(s.stats.stage_ms[k]), 2) for k in range(8)], flush=True)
