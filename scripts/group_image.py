#!/usr/bin/env python3
"""Feasibility count for the gathers of GROUPS of consecutive chunks served from an LDS image staged once (the
fourth design): for groups of GC consecutive chunks of S / St — the union of the columns they touch, merged into runs (gap <= GAP
columns, starts on even columns) — how long is the image, how many runs, and how many doubles are staged against the entries gathered.
usage: group_image.py [scene] [res] [GAP]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polystokes_amd
from polystokes_amd import scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
GAP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc, p = getattr(scenes, scene)(n, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
out = {"scene": "%s%d" % (scene, n), "gap": GAP}
for M in ("S", "St"):
    ptr, col = s.array(M + ".ptr").astype(np.int64), s.array(M + ".col").astype(np.int64)
    ci = s.array(M + ".chunkInfo").reshape(-1, 4)
    row0, rows = ci[:, 2].astype(np.int64), (ci[:, 1].astype(np.int64) >> 16) & 0xffff
    nC = len(row0)
    res = {}
    for GC in (4, 8, 16):
        lens, nruns, ents = [], [], []
        step = max(1, (nC // GC) // 4000)                      # sample of the groups
        for g in range(0, nC // GC, step):
            a, b = g * GC, min(nC, g * GC + GC)
            lo, hi = ptr[row0[a]], ptr[row0[b - 1] + rows[b - 1]]
            u = np.unique(col[lo:hi])
            if u.size == 0:
                continue
            brk = np.nonzero(np.diff(u) > GAP)[0]
            starts = np.concatenate(([u[0]], u[brk + 1])) & ~1
            ends = np.concatenate((u[brk], [u[-1]])) + 1
            L = ((ends - starts + 1) & ~1)
            lens.append(int(L.sum())); nruns.append(len(L)); ents.append(int(hi - lo))
        lens, nruns, ents = np.array(lens), np.array(nruns), np.array(ents)
        res["GC%d" % GC] = {"groups_sampled": int(len(lens)), "image_doubles_mean": float(lens.mean()), "image_doubles_p99": float(np.percentile(lens, 99)), "image_doubles_max": int(lens.max()),
                            "runs_mean": float(nruns.mean()), "runs_p99": float(np.percentile(nruns, 99)), "runs_max": int(nruns.max()),
                            "staged_doubles_per_gathered_entry": float(lens.sum() / ents.sum()),
                            "glds_instr_per_group_mean": float(np.mean([0])),
                            "share_of_groups_with_image_le_2560": float((lens <= 2560).mean()), "le_4096": float((lens <= 4096).mean())}
    out[M] = res
print(json.dumps(out, indent=1))
