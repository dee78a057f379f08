#!/usr/bin/env python3
"""How many lines would St's gathers fetch if the 64-row units of all seven kinds of DOF of the SAME voxels shared one staged copy?
Items = (lattice block, k-plane, group g of 64 rows of every kind's segment); compare with the per-unit sum (what the row-per-lane
kernels fetch today).  usage: group_items.py [scene] [res]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polystokes_amd
from polystokes_amd import _abi as abi, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc, p = getattr(scenes, scene)(n, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
perm = s.array("sysPerm").astype(np.int64)          # reference index -> internal row of St
nsys = perm.size
cidx = s.array("centerActiveIndices").astype(np.int64)
nP = int((cidx >= 0).sum())
shapes = abi.grid_shapes(sc.nx, sc.ny, sc.nz)
kind_of = np.empty(nsys, np.int8); blk = np.empty(nsys, np.int64); pl = np.empty(nsys, np.int8)
nbx, nby = (sc.nx + 16) // 16, (sc.ny + 16) // 16


def place(idx_flat, shape, ref_off, kind):
    idx = idx_flat.reshape(shape)
    kk, jj, ii = np.nonzero(idx >= 0)
    ref = idx[kk, jj, ii].astype(np.int64) + ref_off
    rows = perm[ref]
    kind_of[rows] = kind
    blk[rows] = (ii >> 4) + nbx * ((jj >> 4) + nby * (kk >> 4))
    pl[rows] = kk & 15


for a, off in enumerate((0, nP, 2 * nP, 3 * nP)):
    place(cidx, shapes["center"], off, a)
nE = []
eoff = 4 * nP
for a, nm in enumerate(("edgeYZ", "edgeXZ", "edgeXY")):
    e = s.array(nm + "ActiveIndices").astype(np.int64)
    place(e, shapes[nm], eoff, 4 + a)
    eoff += int((e >= 0).sum())
ptr = s.array("St.ptr").astype(np.int64)
col = s.array("St.col").astype(np.int64)
ln = np.diff(ptr)
nnz = int(col.size)
# rank of every row inside its (block, plane, kind) segment: rows are numbered kind-major inside a plane, so segments are contiguous
key = (blk * 16 + pl) * 8 + kind_of
rows = np.arange(nsys)
order = np.argsort(key, kind="stable")
ks = key[order]
start = np.r_[True, ks[1:] != ks[:-1]]
seg_first = np.maximum.accumulate(np.where(start, np.arange(nsys), 0))
rank = np.empty(nsys, np.int64); rank[order] = np.arange(nsys) - seg_first
item_key = (blk * 16 + pl) * 64 + (rank >> 6)            # all kinds' group g of one plane
unit_key = key * 64 + (rank >> 6)
row_of_entry = np.repeat(rows, ln)
lines = col >> 4


def distinct(keys):
    k = keys[row_of_entry] * (1 << 24) + lines             # (group, line) pairs
    return int(np.unique(k).size)


out = {"scene": scene, "res": n, "nnz": nnz, "vector_lines": int(lines.max()) + 1,
       "unit_lines": distinct(unit_key), "item_lines": distinct(item_key),
       "items": int(np.unique(item_key).size), "units": int(np.unique(unit_key).size)}
ik = item_key[row_of_entry] * (1 << 24) + lines
u = np.unique(ik)
per_item = np.bincount(np.unique(u >> 24, return_inverse=True)[1])
out["item_lines_max"] = int(per_item.max()); out["item_lines_p99"] = int(np.percentile(per_item, 99)); out["item_lines_mean"] = float(per_item.mean())
# exact column segments per item (columns up to GAP apart share one): how many, how many staged columns
ic = np.unique(item_key[row_of_entry] * (1 << 32) + col)
it_, cc = ic >> 32, ic & 0xffffffff
newitem = np.r_[True, it_[1:] != it_[:-1]]
gap = np.r_[0, np.diff(cc)]
nA = int(s.nA)
for G in (8, 16, 32, 64):
    st = newitem | (gap > G)
    segid = np.cumsum(st) - 1
    nseg = np.bincount(np.unique(it_, return_inverse=True)[1], weights=st.astype(np.int64)).astype(np.int64)
    lo = cc[st]; hi = np.r_[cc[1:][st[1:]] * 0, 0]  # placeholder
    last = np.r_[st[1:], True]
    span = cc[last] - cc[st] + 1
    seg_item = it_[st]
    staged = np.bincount(np.unique(seg_item, return_inverse=True)[1], weights=span).astype(np.int64)
    skin = (cc[st] >= nA)
    out["gap%d" % G] = {"segs_mean": float(nseg.mean()), "segs_p99": int(np.percentile(nseg, 99)), "segs_max": int(nseg.max()),
                        "staged_mean": float(staged.mean()), "staged_max": int(staged.max()), "segs_on_skin_rows_mean": float(skin.sum() / nseg.size),
                        "hist": np.bincount(np.minimum(nseg, 63)).tolist()}
print(json.dumps(out))
s.close()
