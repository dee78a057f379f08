#!/usr/bin/env python3
"""Feasibility count for block-resident gathers (last bullet): if the gathered vector of a 16^3 lattice block
(S: the block's DOF range; St: the block's active face rows + its tile's skin rows) sat in LDS, how many gathers would be served from
it, and how many line fills would a launch make (window copies + the distinct lines of the gathers that leave the window, per 64-row
unit) against the sum over units it makes now?   usage: block_window.py [scene] [res]"""
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


def blocks_of(names, dims_list, perm, offsets):
    out = np.full(perm.size, -1, np.int64)
    for kind, (nm, d) in enumerate(zip(names, dims_list)):
        idx = s.array(nm).astype(np.int64).reshape(d[2], d[1], d[0])
        k, j, i = np.nonzero(idx >= 0)
        out[perm[idx[k, j, i] + offsets[kind]]] = ((k >> 4) * 64 + (j >> 4)) * 64 + (i >> 4)
    return out


fd = [(nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1)]
nF = [int((s.array("face%sActiveIndices" % a) >= 0).sum()) for a in "XYZ"]
rowBlk = blocks_of(["face%sActiveIndices" % a for a in "XYZ"], fd, rowPerm, [0, nF[0], nF[0] + nF[1]])
cd = (nx, ny, nz)
ed = [(nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz)]
nE = [int((s.array(e + "ActiveIndices") >= 0).sum()) for e in ("edgeYZ", "edgeXZ", "edgeXY")]
dofBlk = blocks_of(["centerActiveIndices"] * 4 + ["edgeYZActiveIndices", "edgeXZActiveIndices", "edgeXYActiveIndices"], [cd] * 4 + ed, sysPerm,
                   [0, nP, 2 * nP, 3 * nP, 4 * nP, 4 * nP + nE[0], 4 * nP + nE[0] + nE[1]])
# skin rows: the lattice block of the tile's first cell
reg = s.array("reducedRowRegion").astype(np.int64)
face = s.array("reducedRowFace").astype(np.int64)
fi, fj, fk = face & 1023, (face >> 10) & 1023, (face >> 20) & 1023
R = int(reg.max()) + 1 if reg.size else 0
mn = np.full((R, 3), 1 << 30, np.int64)
for a, v in enumerate((fi, fj, fk)):
    np.minimum.at(mn[:, a], reg, v)
regBlk = ((mn[:, 2] >> 4) * 64 + (mn[:, 1] >> 4)) * 64 + (mn[:, 0] >> 4)
rowBlkAll = np.concatenate([rowBlk, regBlk[reg]])


def study(name, rBlk, cBlk):
    ptr = s.array(name + ".ptr").astype(np.int64); col = s.array(name + ".col").astype(np.int64)
    ln = np.diff(ptr)
    erow = np.repeat(np.arange(ln.size), ln)
    inside = rBlk[erow] == cBlk[col]
    unit = erow >> 6                                   # (64 consecutive rows: the chunk cuts move this by a per cent)
    now = np.unique(unit * (1 << 26) + (col >> 4)).size
    out_lines = np.unique((unit * (1 << 26) + (col >> 4))[~inside]).size
    win_lines = int(np.ceil(np.bincount(cBlk, minlength=1).astype(np.float64) / 16).sum())   # every block's window copied once, whole lines
    return {"entries": int(col.size), "inside_window": round(float(inside.mean()), 4), "lines_now_sum_over_units": int(now),
            "lines_with_windows": {"window_copies": win_lines, "gathers_that_leave_the_window": int(out_lines), "total": int(win_lines + out_lines)},
            "ratio": round((win_lines + out_lines) / now, 3)}


print(json.dumps({"scene": sc.name, "S (window = the row's block's DOFs)": study("S", rowBlkAll, dofBlk),
                  "St (window = the DOF's block's face rows + its tile's skin rows)": study("St", dofBlk, rowBlkAll)}, indent=1))
s.close()
