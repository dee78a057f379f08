#!/usr/bin/env python3
"""Which halo samples do a rank's rows touch, by neighbour offset?  (r05: why Dist::decideExchangeMode falls back to the forwarding rounds.)
For every rank of a brick decomposition: the columns of its S outside its owned DOF range, decoded to (sample grid, local i j k) and classed by the
offset of their owner (-1 / 0 / +1 per axis).  usage: diag_halo.py [scene] [res] [dx dy dz]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
scene = sys.argv[1] if len(sys.argv) > 1 else "cavity"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dims = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (2, 2, 2)
sc, p = getattr(scenes, scene)(res, tile=16, pad=2)
g = polystokes_amd.Group(dims[0] * dims[1] * dims[2], dims=dims)
g.solve_scene(sc, p)
GR = ["center"] * 4 + ["edgeYZ", "edgeXZ", "edgeXY"]
ONPLANE = {"center": (0, 0, 0), "edgeYZ": (0, 1, 1), "edgeXZ": (1, 0, 1), "edgeXY": (1, 1, 0)}
for r, b in enumerate(g.bricks):
    s = g.ranks[r]
    nx, ny, nz = b.n_local
    owned = int(s.dist_stats()["owned_dofs"])
    col = np.unique(s.array("S.col").astype(np.int64))
    halo = col[col >= owned]
    sysPerm = s.array("sysPerm").astype(np.int64)
    inv = np.empty(sysPerm.size, np.int64); inv[sysPerm] = np.arange(sysPerm.size)
    ref = inv[halo]                                        # reference-order index of every touched halo DOF
    nP = s.nP
    nE = [int((s.array(e + "ActiveIndices") >= 0).sum()) for e in ("edgeYZ", "edgeXZ", "edgeXY")]
    offs = [0, nP, 2 * nP, 3 * nP, 4 * nP, 4 * nP + nE[0], 4 * nP + nE[0] + nE[1], 4 * nP + sum(nE)]
    shapes = {"center": (nz, ny, nx), "edgeYZ": (nz + 1, ny + 1, nx), "edgeXZ": (nz + 1, ny, nx + 1), "edgeXY": (nz, ny + 1, nx + 1)}
    cnt = collections.Counter()
    for kind in range(7):
        sel = ref[(ref >= offs[kind]) & (ref < offs[kind + 1])] - offs[kind]
        if sel.size == 0:
            continue
        gname = GR[kind]
        idx = s.array(gname + "ActiveIndices").astype(np.int64).reshape(shapes[gname])
        pos = np.full(int(idx.max()) + 1, -1, np.int64)
        kk, jj, ii = np.nonzero(idx >= 0)
        pos[idx[kk, jj, ii]] = np.arange(kk.size)
        q = pos[sel]
        for (i, j, k) in zip(ii[q], jj[q], kk[q]):
            o = []
            for a, c in enumerate((i, j, k)):
                onp = ONPLANE[gname][a]
                if c < b.lo[a]: o.append(-1)
                elif c > b.hi[a] or (c == b.hi[a] and (not onp or b.hasUpper[a])): o.append(+1)
                else: o.append(0)
            cnt[(("p", "txx", "tyy", "tzz", "tyz", "txz", "txy")[kind], tuple(o))] += 1
    diag = {k: v for k, v in cnt.items() if sum(1 for x in k[1] if x) >= 2}
    print("rank", r, "coord", b.coord, "halo DOFs touched", halo.size, "| on diagonal neighbours:", diag if diag else "none", flush=True)
g.close()
