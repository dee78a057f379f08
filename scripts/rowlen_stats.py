#!/usr/bin/env python3
"""Row-length statistics of S and St (padding of a per-wave ELL layout: 64 consecutive rows padded to their longest row).
usage: rowlen_stats.py [scene] [res]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polystokes_amd
from polystokes_amd import _abi as abi, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sc, p = getattr(scenes, scene)(n, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
out = {"scene": scene, "res": n}
nA = int(s.stats.dimData[0]) if hasattr(s.stats, "dimData") else 0
for name in ("S", "St"):
    ptr = s.array(name + ".ptr").astype(np.int64)
    ln = np.diff(ptr)
    rows = ln.size
    pad = (-rows) % 64
    l64 = np.concatenate([ln, np.zeros(pad, np.int64)]).reshape(-1, 64)
    mx = l64.max(axis=1)
    mx2 = (mx + 1) // 2 * 2
    o = {"rows": int(rows), "nnz": int(ln.sum()), "mean": float(ln.mean()), "max": int(ln.max()),
         "hist": np.bincount(ln, minlength=9).tolist(),
         "ell64_slots": int(mx.sum() * 64), "ell64_even_slots": int(mx2.sum() * 64),
         "ell64_over_nnz": float(mx.sum() * 64 / ln.sum()), "wave_width_hist": np.bincount(mx, minlength=9).tolist()}
    # per-wave width with the 4-entry-group form of today: ceil(sum/4)*4 per 256 rows
    out[name] = o
print(json.dumps(out))
s.close()
