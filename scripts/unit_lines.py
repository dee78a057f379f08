#!/usr/bin/env python3
"""Distinct 128-byte lines the 64-row units of the row-per-lane SpMV kernels gather from — as they are cut now (64 consecutive rows of a
chunk) and under other ways of cutting the SAME row sequence into units, from the real matrices in the internal numbering.
The kernels' time follows this count (profiles/r03_spmv_issue.md section 5: L1 -> L2 requests = sum over units of distinct lines).
  groups: an active face row / a DOF belongs to (16^3 lattice block, k-plane, kind); skin rows to (region).
  cuts:   "now"     the chunk table's units
          "group"   a unit never crosses a group boundary (units of <= 64 rows inside a group)
          "group16" the same, and groups shorter than 16 rows are merged with their successor
  cols:   "now" / "aligned" (every (block, plane, kind) group of the gathered vector starts on a 128-byte line: what padding the
          numbering would give)
usage: unit_lines.py [scene] [res]"""
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
nx, ny, nz = sc.nx, sc.ny, sc.nz
nA, nP = s.nA, s.nP
rowPerm, sysPerm = s.array("rowPerm").astype(np.int64), s.array("sysPerm").astype(np.int64)


def groups_of(names, dims_list, perm, offsets):
    """group key per INTERNAL index for the samples of several grids: (block, plane, kind)"""
    out = np.full(perm.size, -1, np.int64)
    for kind, (nm, d) in enumerate(zip(names, dims_list)):
        idx = s.array(nm).astype(np.int64).reshape(d[2], d[1], d[0])
        k, j, i = np.nonzero(idx >= 0)
        ref = idx[k, j, i] + offsets[kind]
        blk = ((k >> 4) * 64 + (j >> 4)) * 64 + (i >> 4)
        out[perm[ref]] = (blk * 16 + (k & 15)) * 8 + kind
    return out


fd = [(nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1)]
nF = [int((s.array("face%sActiveIndices" % a) >= 0).sum()) for a in "XYZ"]
rowGroup = groups_of(["face%sActiveIndices" % a for a in "XYZ"], fd, rowPerm, [0, nF[0], nF[0] + nF[1]])
cd = (nx, ny, nz)
ed = [(nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz)]
nE = [int((s.array(e + "ActiveIndices") >= 0).sum()) for e in ("edgeYZ", "edgeXZ", "edgeXY")]
dofGroup = groups_of(["centerActiveIndices"] * 4 + ["edgeYZActiveIndices", "edgeXZActiveIndices", "edgeXYActiveIndices"], [cd] * 4 + ed, sysPerm,
                     [0, nP, 2 * nP, 3 * nP, 4 * nP, 4 * nP + nE[0], 4 * nP + nE[0] + nE[1]])
assert (rowGroup >= 0).all() and (dofGroup >= 0).all()
reg = s.array("reducedRowRegion").astype(np.int64)
rowGroupAll = np.concatenate([rowGroup, (1 << 40) + reg])          # skin rows: one group per region


def aligned_map(group):
    """new index of every entry when each run of equal group keys starts on a multiple of 16"""
    n = group.size
    start = np.ones(n, bool); start[1:] = group[1:] != group[:-1]
    first = np.nonzero(start)[0]
    size = np.diff(np.append(first, n))
    base = np.concatenate([[0], np.cumsum((size + 15) // 16 * 16)[:-1]])
    gid = np.cumsum(start) - 1
    return base[gid] + (np.arange(n) - first[gid]), int(((size + 15) // 16 * 16).sum())


def units_now(name):
    ci = s.array(name + ".chunkInfo").reshape(-1, 4)
    r0 = ci[:, 2].astype(np.int64); nr = (ci[:, 1].view(np.uint32) >> 16).astype(np.int64)
    u0, ur = [], []
    for w in range(4):
        r = np.clip(nr - 64 * w, 0, 64); m = r > 0
        u0.append(r0[m] + 64 * w); ur.append(r[m])
    u0 = np.concatenate(u0); ur = np.concatenate(ur)
    o = np.argsort(u0, kind="stable")
    return u0[o], ur[o]


def units_cut(group, minrows):
    n = group.size
    start = np.ones(n, bool); start[1:] = group[1:] != group[:-1]
    if minrows > 1:                                   # merge short groups with their successor
        first = np.nonzero(start)[0]
        size = np.diff(np.append(first, n))
        keep = np.ones(first.size, bool)
        acc = 0
        for q in range(first.size):                   # (a few 10^5 groups: fine)
            if acc > 0 and acc < minrows:
                keep[q] = False
            if keep[q]:
                acc = 0
            acc += size[q]
        start[:] = False; start[first[keep]] = True
    first = np.nonzero(start)[0]
    size = np.diff(np.append(first, n))
    u0, ur = [], []
    for f, z in zip(first, size):
        for o in range(0, z, 64):
            u0.append(f + o); ur.append(min(64, z - o))
    return np.array(u0, np.int64), np.array(ur, np.int64)


def lines(ptr, col, u0, ur, colmap=None):
    """sum over units of the distinct lines of the unit's gathered columns; also per-slot instruction count (sum of unit widths)"""
    ln = np.diff(ptr)
    rows_unit = np.zeros(ptr.size - 1, np.int64)
    uid = np.repeat(np.arange(u0.size), ur)
    rows = np.repeat(u0, ur) + (np.arange(ur.sum()) - np.repeat(np.cumsum(ur) - ur, ur))
    rows_unit[rows] = uid
    ent_unit = np.repeat(rows_unit, ln)
    c = col if colmap is None else colmap[col]
    key = ent_unit * (1 << 26) + (c >> 4)
    tot = np.unique(key).size
    wmax = np.zeros(u0.size, np.int64)
    np.maximum.at(wmax, rows_unit, ln)
    return int(tot), int(((wmax + 1) // 2 * 2).sum())


out = {"scene": sc.name}
for name, group, colGroup in (("S", rowGroupAll, dofGroup), ("St", dofGroup, rowGroupAll)):
    ptr = s.array(name + ".ptr").astype(np.int64); col = s.array(name + ".col").astype(np.int64)
    cmap, padded = aligned_map(colGroup)
    res = {"rows": int(ptr.size - 1), "nnz": int(col.size), "row_groups": int(1 + (group[1:] != group[:-1]).sum()), "distinct_row_group_keys": int(np.unique(group).size), "lines_of_vector": int((col.max() + 16) // 16), "vector_padded_to": padded}
    for cut, (u0, ur) in (("now", units_now(name)), ("group", units_cut(group, 1)), ("group16", units_cut(group, 16))):
        for cm_name, cm in (("now", None), ("aligned", cmap)):
            L, slots = lines(ptr, col, u0, ur, cm)
            res["%s/%s" % (cut, cm_name)] = {"units": int(u0.size), "lines": L, "gather_instr_slots": slots}
    out[name] = res
    print(json.dumps({name: res}), flush=True)
s.close()
