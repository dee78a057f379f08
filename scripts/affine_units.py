#!/usr/bin/env python3
"""How regular are the 64-row units of the row-per-lane SpMV streams?  ("affine-rule units")
For every unit (64 consecutive rows of a chunk, one wave) and every slot k (k-th entry of the rows in CSR order): the lanes' columns
form SEGMENTS  col = base + lane  with one value code; a slot with one segment covering all its live lanes needs no per-lane stream
at all (wave-uniform base + code: scalar loads).  Prints, per matrix, the share of units / entries by the largest number of segments
a slot of the unit needs, and the distinct unit descriptors (segment tables relative to the unit's first column) — what a
descriptor table shared between equivalent units would hold.
usage: affine_units.py [scene] [res]     (one JSON line per matrix)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polystokes_amd
from polystokes_amd import scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc, p = getattr(scenes, scene)(n, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()


def study(name):
    ptr = s.array(name + ".ptr").astype(np.int64)
    col = s.array(name + ".col").astype(np.int64)
    code = s.array(name + ".code").astype(np.int64)
    ci = s.array(name + ".chunkInfo").reshape(-1, 4)
    rows0 = ci[:, 2].astype(np.int64)
    nrows = (ci[:, 1].view(np.uint32) >> 16).astype(np.int64)
    # unit table: (first row, rows)
    u0, ur = [], []
    for w in range(4):
        r = np.clip(nrows - 64 * w, 0, 64)
        m = r > 0
        u0.append(rows0[m] + 64 * w); ur.append(r[m])
    u0 = np.concatenate(u0); ur = np.concatenate(ur)
    order = np.argsort(u0, kind="stable"); u0 = u0[order]; ur = ur[order]
    U = u0.size
    ln = np.diff(ptr)
    # dense (U, 64, 8) tables of columns / codes, -1 = no entry
    lane = np.arange(64)
    rowid = u0[:, None] + lane[None, :]
    live = lane[None, :] < ur[:, None]
    rowid = np.where(live, rowid, 0)
    L = np.where(live, ln[rowid], 0)
    W = 8
    C = np.full((U, 64, W), -1, np.int64); V = np.zeros((U, 64, W), np.int64)
    for k in range(W):
        m = L > k
        idx = ptr[rowid] + k
        C[:, :, k] = np.where(m, col[np.where(m, idx, 0)], -1)
        V[:, :, k] = np.where(m, code[np.where(m, idx, 0)], 0)
    has = C >= 0
    rel = C - lane[None, :, None]                      # col - lane: constant inside a segment
    # a new segment starts at lane l if the slot has an entry there and (no entry at l-1, or rel / code differ)
    start = has.copy()
    same = has[:, 1:, :] & has[:, :-1, :] & (rel[:, 1:, :] == rel[:, :-1, :]) & (V[:, 1:, :] == V[:, :-1, :])
    start[:, 1:, :] &= ~same
    segs = start.sum(axis=1)                           # (U, W) segments per slot
    # gaps: lanes without an entry between lanes with (the slot is not one contiguous lane range)
    maxseg = segs.max(axis=1)
    ent = has.sum(axis=(1, 2))
    tot = int(ent.sum())
    out = {"matrix": name, "scene": sc.name, "units": int(U), "entries": tot}
    hist = {}
    for b in (1, 2, 3, 4, 6, 8, 16, 64):
        m = maxseg <= b
        hist[f"<= {b}"] = {"units": round(float(m.mean()), 4), "entries": round(float(ent[m].sum()) / tot, 4)}
    out["units_by_max_segments_per_slot"] = hist
    out["mean_segments_per_unit"] = round(float(segs.sum(axis=1).mean()), 2)
    out["segments_total"] = int(segs.sum())
    # whole-unit regularity: every slot one segment over ALL live lanes
    full = (segs <= 1).all(axis=1) & ((has.sum(axis=1) == ur[:, None]) | (has.sum(axis=1) == 0)).all(axis=1)
    out["fully_affine_units"] = {"units": round(float(full.mean()), 4), "entries": round(float(ent[full].sum()) / tot, 4)}
    # distinct descriptors: the segment table relative to the unit's smallest column
    import hashlib
    base = np.where(has, C, np.iinfo(np.int64).max).min(axis=(1, 2))
    keys = set()
    relu = np.where(has, C - base[:, None, None], -1).astype(np.int32)
    for u in range(U):
        keys.add(hashlib.sha1(relu[u].tobytes() + V[u].astype(np.int8).tobytes()).digest())
    out["distinct_unit_patterns"] = len(keys)
    # lines a unit touches (8-byte entries, 128-byte lines) against its ideal
    lines = np.where(has, C >> 4, -1)
    dl = 0
    for u in range(0, U, 4096):
        blk = lines[u:u + 4096].reshape(min(4096, U - u), -1)
        blk = np.sort(blk, axis=1)
        dl += int(((blk[:, 1:] != blk[:, :-1]) & (blk[:, 1:] >= 0)).sum() + (blk[:, 0] >= 0).sum())
    out["distinct_lines_per_unit_sum"] = dl
    out["lines_of_vector"] = int((col.max() + 16) // 16)
    return out


for nm in ("S", "St"):
    print(json.dumps(study(nm)), flush=True)
s.close()
