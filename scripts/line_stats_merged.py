#!/usr/bin/env python3
"""What would ONE spatial row sequence (active and skin rows interleaved by lattice block / voxel / axis) buy?
Ground truth at G = 16 chunks (the sharing domain the hardware shows): distinct x lines gathered by S and distinct t lines
gathered by St, current row order vs merged order.  cavity N^3."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import _abi as abi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sc, p = scenes.cavity(n, tile=16, pad=2, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0); s.upload(sc, p); s.setup()
nA = s.nA
Sp, Sc = s.array("S.ptr").astype(np.int64), s.array("S.col").astype(np.int64)
Tp, Tc = s.array("St.ptr").astype(np.int64), s.array("St.col").astype(np.int64)
rows = len(Sp) - 1; ndof = len(Tp) - 1
# spatial key of every row: (block, voxel, axis) of its face position
key = np.zeros(rows, np.int64)
sh = abi.grid_shapes(n, n, n)
for a, nm in enumerate("XYZ"):
    fr = s.array("faceRow" + nm).reshape(sh["face" + nm])
    k, j, i = np.nonzero(fr >= 0)
    r = fr[k, j, i]
    blk = (k // 16) * 1000000 + (j // 16) * 1000 + (i // 16)
    vox = (k % 16) * 256 + (j % 16) * 16 + (i % 16)
    key[r] = (blk * 4096 + vox) * 3 + a
order = np.argsort(key, kind="stable")           # new position -> old row
newpos = np.empty(rows, np.int64); newpos[order] = np.arange(rows)
def cost(rowids, cols, ncols, G=16):
    grp = rowids // (256 * G)
    return len(np.unique(grp * (ncols // 16 + 2) + cols // 16)) / (ncols / 16)
rowofS = np.repeat(np.arange(rows), np.diff(Sp))
rowofT = np.repeat(np.arange(ndof), np.diff(Tp))
print("S: distinct x lines / all x lines at G=16   current %.2f   merged %.2f" % (cost(rowofS, Sc, ndof), cost(newpos[rowofS], Sc, ndof)))
print("St: distinct t lines / all t lines at G=16  current %.2f   merged %.2f" % (cost(rowofT, Tc, rows), cost(rowofT, newpos[Tc], rows)))
for G in (4, 64):
    print("G=%d  S %.2f -> %.2f   St %.2f -> %.2f" % (G, cost(rowofS, Sc, ndof, G), cost(newpos[rowofS], Sc, ndof, G), cost(rowofT, Tc, rows, G), cost(rowofT, newpos[Tc], rows, G)))
