"""Where do a brick rank's labels differ from the single domain's?  (fuzz_bricks case by seed: exchange lists that disagree)
usage: brick_diag.py <seed>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import polystokes_amd
from polystokes_amd import partition, _abi as abi
from helpers import fuzz_brick_case
seed = int(sys.argv[1])
sc, p, dims, n, tile = fuzz_brick_case(seed)
print("grid", n, "dims", dims, "tile", tile, "pad", p.tilePadding, "L", p.activeLiquidBoundaryLayerSize, "S", p.activeSolidBoundaryLayerSize)
single = polystokes_amd.Solver(0); single.upload(sc, p); single.setup()
names = ["center", "faceX", "faceY", "faceZ", "edgeYZ", "edgeXZ", "edgeXY"]
ext = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0)]
G = {}
for nm, e in zip(names, ext):
    for kind in ("Labels", "ReducedIndices"):
        G[nm + kind] = single.array(nm + kind).reshape(sc.nz + e[2], sc.ny + e[1], sc.nx + e[0])
world = dims[0] * dims[1] * dims[2]
for r in range(world):
    b = partition.make_brick((sc.nx, sc.ny, sc.nz), dims, r, p.tileSize)
    s = polystokes_amd.Solver(0)
    s.upload(partition.local_scene_brick(sc, b), p); s.set_brick(b)
    # the local setup alone (no exchange): through the group API's first stage is not exposed, so use a 1-rank view: labels come from ps_setup on the local grid
    s.L.ps_set_brick  # (keep the brick: gOff matters for nothing in the labels)
    try:
        s.slabless = True
        st = polystokes_amd.Stats()
        # ps_setup_device refuses a context with a brick: clear it by re-uploading, the labels do not depend on it
        s.upload(partition.local_scene_brick(sc, b), p)
        s.setup()
    except Exception as e:
        print("rank", r, "setup failed", e); continue
    nx, ny, nz = b.n_local
    ox, oy, oz = b.origin
    print("rank", r, "origin", b.origin, "local", b.n_local, "owned lo/hi", b.lo, b.hi)
    for nm, e in zip(names, ext):
        loc = s.array(nm + "Labels").reshape(nz + e[2], ny + e[1], nx + e[0])
        glo = G[nm + "Labels"][oz:oz + nz + e[2], oy:oy + ny + e[1], ox:ox + nx + e[0]]
        # compare on the owned box widened by one sample on every side that has a neighbour
        sl = []
        for a, (lo, hi, nl, ee) in enumerate(zip(b.lo, b.hi, b.n_local, e)):
            a0 = max(lo - 1, 0); a1 = min(hi + 1 + ee, nl + ee)
            sl.append(slice(a0, a1))
        sub_l, sub_g = loc[sl[2], sl[1], sl[0]], glo[sl[2], sl[1], sl[0]]
        d = np.argwhere(sub_l != sub_g)
        if d.size:
            print("   %s labels differ at %d samples (owned box +-1); first (local k,j,i):" % (nm, d.shape[0]),
                  [(int(k + sl[2].start), int(j + sl[1].start), int(i + sl[0].start), int(sub_l[k, j, i]), int(sub_g[k, j, i])) for k, j, i in d[:6]])
    s.close()
single.close()

# ---- detail for one rank: the regions around the first differing cell -------------------------------------------------------------
if len(sys.argv) > 2:
    r = int(sys.argv[2])
    b = partition.make_brick((sc.nx, sc.ny, sc.nz), dims, r, p.tileSize)
    single = polystokes_amd.Solver(0); single.upload(sc, p); single.setup()
    s = polystokes_amd.Solver(0); s.upload(partition.local_scene_brick(sc, b), p); s.setup()
    nx, ny, nz = b.n_local; ox, oy, oz = b.origin
    gl = single.array("centerLabels").reshape(sc.nz, sc.ny, sc.nx); gr = single.array("centerReducedIndices").reshape(sc.nz, sc.ny, sc.nx)
    ll = s.array("centerLabels").reshape(nz, ny, nx); lr = s.array("centerReducedIndices").reshape(nz, ny, nx)
    sub_g = gl[oz:oz + nz, oy:oy + ny, ox:ox + nx]; sub_gr = gr[oz:oz + nz, oy:oy + ny, ox:ox + nx]
    d = np.argwhere(ll != sub_g)
    print("rank", r, "ALL differing cells in the local grid:", d.shape[0], " x range of the differences (global):", (d[:, 2].min() + ox, d[:, 2].max() + ox) if d.size else None)
    def box(mask):
        q = np.argwhere(mask)
        return None if not q.size else [(int(q[:, a].min()), int(q[:, a].max())) for a in (2, 1, 0)]
    seen = set()
    for k, j, i in d[:400]:
        a, c = int(lr[k, j, i]), int(sub_gr[k, j, i])
        if (a, c) in seen: continue
        seen.add((a, c))
        print("  cell local", (int(i), int(j), int(k)), "global", (int(i) + ox, int(j) + oy, int(k) + oz), "label local/global", int(ll[k, j, i]), int(sub_g[k, j, i]), "region local/global", a, c,
              "| local region box (local x,y,z)", box(lr == a) if a >= 0 else None, "cells", int((lr == a).sum()) if a >= 0 else 0,
              "| global region box (global)", box(gr == c) if c >= 0 else None, "cells", int((gr == c).sum()) if c >= 0 else 0)
