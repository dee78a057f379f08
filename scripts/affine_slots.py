#!/usr/bin/env python3
"""How much of the row-per-lane stream is 'affine'?  For every 64-row unit and slot k: do the rows that have an entry k gather
CONSECUTIVE columns (col = base + lane) with ONE value?  Such a slot needs 4 bytes, not 64 x 3.   usage: affine_slots.py [scene] [res]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc, p = getattr(scenes, scene)(n, tile=16, pad=2)
s = polystokes_amd.Solver(0); s.upload(sc, p); s.setup()
out = {"scene": scene, "res": n}
for name, width in (("S", 8), ("St", 6)):
    ptr = s.array(name + ".ptr").astype(np.int64); col = s.array(name + ".col").astype(np.int64); val = s.array(name + ".val")
    ln = np.diff(ptr); rows = ln.size; pad = (-rows) % 64
    o = {"rows": int(rows), "nnz": int(col.size)}
    lane = np.arange(rows + pad) % 64
    tot_slots = 0; reg_entries = 0; reg_slots = 0; piece2 = 0; colreg_entries = 0
    for k in range(width):
        has = np.r_[ln > k, np.zeros(pad, bool)]
        idx = np.where(has[:rows], ptr[:-1] + k, 0)
        c = np.r_[np.where(has[:rows], col[idx], 0), np.zeros(pad, np.int64)] - lane
        v = np.r_[np.where(has[:rows], val[idx], 0.), np.zeros(pad)]
        H = has.reshape(-1, 64); C = c.reshape(-1, 64); V = v.reshape(-1, 64)
        cnt = H.sum(1)
        big = np.int64(1) << 60
        cmin = np.where(H, C, big).min(1); cmax = np.where(H, C, -big).max(1)
        vmin = np.where(H, V, np.inf).min(1); vmax = np.where(H, V, -np.inf).max(1)
        used = cnt > 0
        creg = used & (cmin == cmax)
        reg = creg & (vmin == vmax)
        tot_slots += int(used.sum()); reg_slots += int(reg.sum()); reg_entries += int(cnt[reg].sum()); colreg_entries += int(cnt[creg].sum())
    o["unit_slots"] = tot_slots; o["affine_slots"] = reg_slots; o["affine_entries_frac"] = reg_entries / col.size; o["affine_cols_only_entries_frac"] = colreg_entries / col.size
    out[name] = o
print(json.dumps(out))
s.close()
