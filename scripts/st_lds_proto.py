#!/usr/bin/env python3
"""VERDICT r04 item 1, gate for step (B): the St product with every gather from an LDS image staged once per lattice block
(scripts/t_st_lds.hip).  numpy builds, from the library's own St (CSR in the internal numbering), per 16^3 lattice block:
  the image  = [zero pair][the block's active face rows, one range of t][the skin rows of its tile, one range][halo pairs: list]
  the stream = 16-bit image positions + the int8 value codes, row-per-lane units as the product has them; blocks with identical
               streams share one
then checks the kernel against a scipy product and times it next to the product's own St kernel in the same process.
Counts for the gate (A) of the verdict go to gpurun_out/<tag>.json: image doubles per block, halo share, staged doubles per gathered entry.
usage: st_lds_proto.py [res] [tag] [scene]"""
import ctypes as C
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tag = sys.argv[2] if len(sys.argv) > 2 else "st_lds_%d" % res
scene = sys.argv[3] if len(sys.argv) > 3 else "cavity"
so = os.path.join(ROOT, "scripts", "_bin", "libt_st_lds.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, "scripts", "t_st_lds.hip")):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-w",
                           os.path.join(ROOT, "scripts", "t_st_lds.hip"), "-o", so])
import torch
torch.cuda.init()                                          # torch's HIP runtime first: it finds no device once the library has initialised its own
import polystokes_amd
from polystokes_amd import scenes
T0 = time.time()
CAP = int(os.environ.get("ST_LDS_CAP", "9728"))            # doubles per image: two workgroups per CU


def log(*a):
    print("[%6.1f s]" % (time.time() - T0), *a, flush=True)


sc, p = getattr(scenes, scene)(res, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
nx, ny, nz = sc.nx, sc.ny, sc.nz
nA, nP = s.nA, s.nP
which = sys.argv[4] if len(sys.argv) > 4 else "St"          # St: rows = DOFs, image of t;  S: rows = face rows, image of p
ptr = s.array(which + ".ptr").astype(np.int64)
col = s.array(which + ".col").astype(np.int64)
code = s.array(which + ".code").astype(np.int8)
rows = ptr.size - 1
nnz = col.size
tLen = int(col.max()) + 1
log("%s: rows %d nnz %d cols %d (nA %d)" % (which, rows, nnz, tLen, nA))
sysPerm, rowPerm = s.array("sysPerm").astype(np.int64), s.array("rowPerm").astype(np.int64)
nbx, nby = (nx + 16) >> 4, (ny + 16) >> 4          # (edge / face grids reach index n: one more block per axis is harmless)


def block_of(names, dims_list, perm, offsets, n_out):
    out = np.full(n_out, -1, np.int64)
    for kind, (nm, d) in enumerate(zip(names, dims_list)):
        idx = s.array(nm).astype(np.int64).reshape(d[2], d[1], d[0])
        k, j, i = np.nonzero(idx >= 0)
        out[perm[idx[k, j, i] + offsets[kind]]] = ((k >> 4) * nby + (j >> 4)) * nbx + (i >> 4)
    return out


fd = [(nx + 1, ny, nz), (nx, ny + 1, nz), (nx, ny, nz + 1)]
nF = [int((s.array("face%sActiveIndices" % a) >= 0).sum()) for a in "XYZ"]
faceBlk = block_of(["face%sActiveIndices" % a for a in "XYZ"], fd, rowPerm, [0, nF[0], nF[0] + nF[1]], nA)
cd = (nx, ny, nz)
ed = [(nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz)]
nE = [int((s.array(e + "ActiveIndices") >= 0).sum()) for e in ("edgeYZ", "edgeXZ", "edgeXY")]
nDof = 4 * nP + sum(nE)
dofBlk = block_of(["centerActiveIndices"] * 4 + ["edgeYZActiveIndices", "edgeXZActiveIndices", "edgeXYActiveIndices"], [cd] * 4 + ed, sysPerm,
                  [0, nP, 2 * nP, 3 * nP, 4 * nP, 4 * nP + nE[0], 4 * nP + nE[0] + nE[1]], nDof)
assert (faceBlk >= 0).all() and (dofBlk >= 0).all()
assert (np.diff(dofBlk) >= 0).all() and (np.diff(faceBlk) >= 0).all(), "the internal numbering is not block-major"
reg = s.array("reducedRowRegion").astype(np.int64)
assert (np.diff(reg) >= 0).all()
nReg = int(reg.max()) + 1 if reg.size else 0
regPtr = np.searchsorted(reg, np.arange(nReg + 1), "left")
lens = np.diff(ptr)


def most_frequent(keyA, keyB, nAkeys):
    """for every value of keyA (0 .. nAkeys - 1): the value of keyB that accompanies it most often (-1: never seen)"""
    m = int(keyB.max()) + 2 if keyB.size else 2
    uk, cnt = np.unique(keyA * m + keyB, return_counts=True)
    ua, ubv = uk // m, uk % m
    best_ = np.full(nAkeys, -1, np.int64)
    order = np.lexsort((cnt, ua))
    lastOf = np.flatnonzero(np.concatenate((ua[order][1:] != ua[order][:-1], [True]))) if ua.size else np.zeros(0, np.int64)
    best_[ua[order][lastOf]] = ubv[order][lastOf]
    return best_


# HOME lattice block of every row; a "block" below = a run of consecutive rows with one home
if which == "St":
    rowHome = dofBlk
else:
    rowOfNz0 = np.repeat(np.arange(rows), lens)
    sk = rowOfNz0 >= nA
    regHome = most_frequent(reg[rowOfNz0[sk] - nA], dofBlk[col[sk]], nReg)
    rowHome = np.concatenate((faceBlk, regHome[reg]))
    del rowOfNz0, sk
bstart = np.concatenate(([0], np.flatnonzero(np.diff(rowHome)) + 1))
bid = rowHome[bstart]
nBlk = bstart.size
bd0 = bstart
bd1 = np.concatenate((bstart[1:], [rows]))
rowBlk = np.repeat(np.arange(nBlk), bd1 - bd0)            # run ("block", dense id) of every row
nzBlk = np.repeat(rowBlk, lens)                            # ... of every entry
if which == "St":
    # range 1: the active face rows of the home block; range 2: the skin rows of its tile = the region its rows reference most
    ba0 = np.searchsorted(faceBlk, bid, "left")
    ba1 = np.searchsorted(faceBlk, bid, "right")
    skin = col >= nA
    best = most_frequent(nzBlk[skin], reg[col[skin] - nA], nBlk)
    bs0 = np.where(best >= 0, nA + regPtr[np.maximum(best, 0)], 0)
    bs1 = np.where(best >= 0, nA + regPtr[np.maximum(best, 0) + 1], 0)
    del skin
else:
    # range 1: the DOFs of the home block; no second range
    ba0 = np.searchsorted(dofBlk, bid, "left")
    ba1 = np.searchsorted(dofBlk, bid, "right")
    bs0 = np.zeros(nBlk, np.int64); bs1 = np.zeros(nBlk, np.int64)
inAblk = (col >= ba0[nzBlk]) & (col < ba1[nzBlk])
inSblk = (col >= bs0[nzBlk]) & (col < bs1[nzBlk])
if which == "S":
    # skin rows are ordered (region, length class, axis, position): any part of them touches DOFs all over the block — their image is the
    # LIST of the 16-byte pairs they touch, no range
    inAblk &= (bd0 < nA)[nzBlk]
# ITEMS: a block, or — where the image of a whole block does not fit — a block's units dealt evenly to K parts.  Per item the two
# ranges are the tight [min, max] of the columns its rows touch inside the block's active rows / its tile's skin rows.
relB = np.arange(rows) - bd0[rowBlk]
nUb = (bd1 - bd0 + 63) // 64
K = np.ones(nBlk, np.int64)
BIG = np.int64(1) << 40
for attempt in range(8):
    itemStart = np.concatenate(([0], np.cumsum(K)))
    rowB = itemStart[rowBlk] + ((relB >> 6) * K[rowBlk]) // nUb[rowBlk]
    nB = int(itemStart[-1])
    first = np.flatnonzero(np.concatenate(([True], rowB[1:] != rowB[:-1])))
    assert first.size == nB
    d0 = first
    d1 = np.concatenate((first[1:], [rows]))
    nzB = np.repeat(rowB, lens)
    nzFirst = ptr[d0]
    okI = ptr[d1] > nzFirst                                 # items with entries
    def red(fn, arr, fill):
        out = np.full(nB, fill, np.int64)
        out[okI] = fn.reduceat(arr, nzFirst[okI])
        return out
    aMin = red(np.minimum, np.where(inAblk, col, BIG), BIG); aMax = red(np.maximum, np.where(inAblk, col, -1), -1)
    sMin = red(np.minimum, np.where(inSblk, col, BIG), BIG); sMax = red(np.maximum, np.where(inSblk, col, -1), -1)
    a0e = np.where(aMax >= 0, aMin & ~1, 0); lenA = np.where(aMax >= 0, ((aMax + 1 - a0e + 127) // 128) * 128, 0)
    s0e = np.where(sMax >= 0, sMin & ~1, 0); lenS = np.where(sMax >= 0, ((sMax + 1 - s0e + 127) // 128) * 128, 0)
    inA = (col >= a0e[nzB]) & (col < (a0e + lenA)[nzB])
    inS = (~inA) & (col >= s0e[nzB]) & (col < (s0e + lenS)[nzB])
    rest = ~(inA | inS)
    hkey = (nzB[rest] << 32) | (col[rest] >> 1)
    hu, hinv = np.unique(hkey, return_inverse=True)
    hb = hu >> 32
    haloBase = np.searchsorted(hb, np.arange(nB), "left")
    nPairs = np.searchsorted(hb, np.arange(nB), "right") - haloBase
    img = 2 + lenA + lenS + ((nPairs + 63) // 64) * 128
    over = img > CAP
    log("attempt %d: %d items of %d blocks, %d over the cap of %d doubles (max %d)" % (attempt, nB, nBlk, int(over.sum()), CAP, int(img.max())))
    if not over.any():
        break
    blkOfItem = np.repeat(np.arange(nBlk), K)
    grow = np.zeros(nBlk, bool); grow[blkOfItem[over]] = True
    K = np.where(grow, np.minimum(K + 1, nUb), K)
assert not over.any()
haloPairs = (hu & 0xffffffff).astype(np.int32)
pos = np.empty(nnz, np.int64)
pos[inA] = 2 + col[inA] - a0e[nzB[inA]]
pos[inS] = 2 + lenA[nzB[inS]] + col[inS] - s0e[nzB[inS]]
b_r = nzB[rest]
pos[rest] = 2 + lenA[b_r] + lenS[b_r] + 2 * (hinv - haloBase[b_r]) + (col[rest] & 1)
log("items %d (blocks %d); image doubles per item: mean %.0f p99 %.0f max %d; halo pairs mean %.0f max %d; in A %.3f in S %.3f halo %.3f of the entries"
    % (nB, nBlk, img.mean(), np.percentile(img, 99), img.max(), nPairs.mean(), nPairs.max(), inA.mean(), inS.mean(), rest.mean()))
assert pos.max() < 65536
counts = {"scene": "%s%d" % (scene, res), "blocks": int(nBlk), "items": int(nB), "cap_doubles": CAP, "rows": int(rows), "nnz": int(nnz), "image_doubles_mean": float(img.mean()),
          "image_doubles_p99": float(np.percentile(img, 99)),
          "image_doubles_max": int(img.max()), "halo_pairs_mean": float(nPairs.mean()), "halo_pairs_max": int(nPairs.max()),
          "entries_from_active_range": float(inA.mean()), "entries_from_skin_range": float(inS.mean()), "entries_from_halo": float(rest.mean()),
          "staged_doubles_per_gathered_entry": float(img.sum() / nnz), "staged_doubles_over_t_entries": float(img.sum() / tLen),
          "dma_pieces_per_item_mean": float((img / 128).mean())}
# ---- row-per-lane units (64 rows, never across a block), even width = the unit's longest row
rel = np.arange(rows) - d0[rowB]
nU = (d1 - d0 + 63) // 64
uStart = np.concatenate(([0], np.cumsum(nU)))
unitOfRow = uStart[rowB] + (rel >> 6)
totU = int(uStart[-1])
firstRowOfUnit = np.flatnonzero(np.concatenate(([True], unitOfRow[1:] != unitOfRow[:-1])))
W = np.maximum.reduceat(lens, firstRowOfUnit)
W = (W + 1) & ~1
assert W.max() <= 8
rowsInUnit = np.diff(np.concatenate((firstRowOfUnit, [rows])))
colLen = W * 64                                            # uint16 entries
codeLen = np.where(W > 4, 512, np.where(W > 0, 256, 0))   # bytes
colOff = np.concatenate(([0], np.cumsum(colLen)))
codeOff = np.concatenate(([0], np.cumsum(codeLen)))
ecol = np.zeros(int(colOff[-1]), np.uint16)
ecode = np.zeros(int(codeOff[-1]), np.int8)
rowOfNz = np.repeat(np.arange(rows), lens)
kk = np.arange(nnz) - ptr[rowOfNz]
uu = unitOfRow[rowOfNz]
ln = rel[rowOfNz] & 63
ecol[colOff[uu] + ln * W[uu] + kk] = pos.astype(np.uint16)
ecode[codeOff[uu] + ln * np.where(W[uu] > 4, 8, 4) + kk] = code
log("units %d; column slots %d (%.3f of nnz)" % (totU, ecol.size, ecol.size / nnz))
# ---- blocks with identical streams share one: signature per block, first block of a class is its representative
cb0, cb1 = colOff[uStart[:-1]], colOff[uStart[1:]]
relc = np.arange(ecol.size, dtype=np.uint64) - np.repeat(cb0.astype(np.uint64), (cb1 - cb0))
wgt = (relc * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
nonempty = cb1 > cb0
sig = np.zeros(nB, np.uint64)
sig[nonempty] = np.add.reduceat((ecol.astype(np.uint64) + np.uint64(1)) * wgt, cb0[nonempty])
kb0, kb1 = codeOff[uStart[:-1]], codeOff[uStart[1:]]
relk = np.arange(ecode.size, dtype=np.uint64) - np.repeat(kb0.astype(np.uint64), (kb1 - kb0))
wgk = (relk * np.uint64(0xC2B2AE3D27D4EB4F)) | np.uint64(1)
sig2 = np.zeros(nB, np.uint64)
ne2 = kb1 > kb0
sig2[ne2] = np.add.reduceat((ecode.astype(np.int64) + 129).astype(np.uint64) * wgk, kb0[ne2])
full = np.stack([sig, sig2, (d1 - d0).astype(np.uint64), (cb1 - cb0).astype(np.uint64), (kb1 - kb0).astype(np.uint64)], 1)
_, repIdx, cls = np.unique(full, axis=0, return_index=True, return_inverse=True)
cls = cls.reshape(-1)
rep = repIdx[cls]                                          # representative block of every block
# verify a sample of the classes byte for byte
rng = np.random.default_rng(1)
for b in rng.choice(nB, size=min(nB, 200), replace=False):
    r_ = rep[b]
    assert np.array_equal(ecol[cb0[b]:cb1[b]], ecol[cb0[r_]:cb1[r_]]) and np.array_equal(ecode[kb0[b]:kb1[b]], ecode[kb0[r_]:kb1[r_]]) and np.array_equal(W[uStart[b]:uStart[b + 1]], W[uStart[r_]:uStart[r_ + 1]])
reps = np.unique(rep)
log("stream classes: %d of %d blocks" % (reps.size, nB))
# compact the streams to the representatives
newCol0 = np.zeros(nB, np.int64); newCode0 = np.zeros(nB, np.int64); newU0 = np.zeros(nB, np.int64)
cc = ck = cu = 0
colParts, codeParts, unitParts = [], [], []
for r_ in reps:
    newCol0[r_], newCode0[r_], newU0[r_] = cc, ck, cu
    colParts.append(ecol[cb0[r_]:cb1[r_]]); codeParts.append(ecode[kb0[r_]:kb1[r_]])
    u0_, u1_ = uStart[r_], uStart[r_ + 1]
    un = np.zeros((u1_ - u0_, 4), np.int32)
    un[:, 0] = (colOff[u0_:u1_] - cb0[r_]) * 2              # byte offsets inside the class's stream
    un[:, 1] = codeOff[u0_:u1_] - kb0[r_]
    un[:, 2] = np.arange(u1_ - u0_) * 64                    # first row, relative to the block's
    un[:, 3] = (W[u0_:u1_] << 8) | rowsInUnit[u0_:u1_]
    unitParts.append(un)
    cc += cb1[r_] - cb0[r_]; ck += kb1[r_] - kb0[r_]; cu += u1_ - u0_
ecolC = np.concatenate(colParts); ecodeC = np.concatenate(codeParts); unitsC = np.concatenate(unitParts)
# rows-in-unit of a block may differ from its representative's only if the row counts differ (they are in the signature)
desc = np.zeros((nB, 12), np.int32)
desc[:, 0] = d0; desc[:, 1] = nU; desc[:, 2] = newU0[rep]; desc[:, 3] = newCol0[rep] * 2; desc[:, 4] = newCode0[rep]
desc[:, 5] = a0e; desc[:, 6] = lenA; desc[:, 7] = s0e; desc[:, 8] = lenS; desc[:, 9] = haloBase; desc[:, 10] = nPairs
counts.update({"stream_classes": int(reps.size), "stream_bytes_distinct": int(ecolC.nbytes + ecodeC.nbytes), "stream_bytes_all": int(ecol.nbytes + ecode.nbytes),
               "halo_list_bytes": int(haloPairs.nbytes)})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(counts, open(os.path.join(ROOT, "gpurun_out", tag + "_counts.json"), "w"), indent=1)
log(json.dumps(counts))

# ---- run it
dev = "cuda"
gen = np.random.default_rng(7)
xvec = np.zeros(tLen + 256); xvec[:tLen] = gen.standard_normal(tLen)        # the gathered vector (t for St, p for S)
pvec = gen.standard_normal(rows); rvec = gen.standard_normal(rows)
ucode = gen.integers(0, 256, rows).astype(np.uint8); udict = gen.random(256)
dinv = gen.random(rows).astype(np.float32)
scale, alpha, dtv = 0.5, 0.37, 0.01
import scipy.sparse as sp
Mx = sp.csr_matrix((code.astype(np.float64) * scale, col, ptr), shape=(rows, tLen)) @ xvec[:tLen]
if which == "St":
    rexp = rvec - alpha * (-Mx - 0.5 * udict[ucode] * pvec)
else:
    rexp = np.where(np.arange(rows) < nA, dtv * udict[ucode] * Mx, Mx)
L = C.CDLL(so)
L.st_lds_launch.restype = C.c_int
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
T = dict(desc=tt(desc), units=tt(unitsC), ecol=tt(ecolC.view(np.int16)), ecode=tt(ecodeC), halo=tt(haloPairs if haloPairs.size else np.zeros(1, np.int32)), t=tt(xvec), p=tt(pvec),
         uc=tt(ucode), ud=tt(udict), dinv=tt(dinv))
ldsBytes = int(((img.max() * 8 + 1023) // 1024) * 1024)
results = {"lds_bytes": ldsBytes, "matrix": which}
MODE = 0 if which == "St" else 1


def launch(stage, nu, grid, reps_):
    r = tt(rvec)
    part = torch.zeros(2 * grid, dtype=torch.float64, device=dev)
    ms = C.c_float(0)
    rc = L.st_lds_launch(C.c_int(stage + 4 * MODE + (8 if nu == 4 else 0)), C.c_int(grid), C.c_int(ldsBytes), C.c_void_p(T["desc"].data_ptr()), C.c_int(nB), C.c_void_p(T["units"].data_ptr()),
                         C.c_void_p(T["ecol"].data_ptr()), C.c_void_p(T["ecode"].data_ptr()), C.c_uint(ecolC.nbytes), C.c_uint(ecodeC.nbytes), C.c_void_p(T["halo"].data_ptr()),
                         C.c_void_p(T["t"].data_ptr()), C.c_int(nA if MODE == 1 else tLen), C.c_void_p(T["p"].data_ptr()), C.c_void_p(T["uc"].data_ptr()), C.c_void_p(T["ud"].data_ptr()),
                         C.c_void_p(r.data_ptr()), C.c_void_p(T["dinv"].data_ptr()), C.c_int(rows), C.c_double(scale), C.c_double(dtv if MODE == 1 else alpha), C.c_void_p(part.data_ptr()),
                         C.c_int(reps_), C.byref(ms))
    assert rc == 0, rc
    return r.cpu().numpy(), part.cpu().numpy(), ms.value


VARIANTS = [(0, 2, (512,)), (0, 4, (512,)), (1, 2, (512,)), (2, 2, (256,)), (2, 4, (256,))]
for stage, nu, grids in VARIANTS:
    rgot, part, _ = launch(stage, nu, grids[0], 1)
    err = np.abs(rgot - rexp).max() / np.abs(rexp).max()
    log("stage %d, %d units in flight: max rel err %.2e" % (stage, nu, err))
    assert err < 1e-12
    results["err_stage%d_nu%d" % (stage, nu)] = float(err)
for rnd in range(2):
    for stage, nu, grids in VARIANTS:
        for grid in grids:
            _, _, ms = launch(stage, nu, grid, 20)
            log("stage %d, %d units in flight, grid %4d: %.4f ms" % (stage, nu, grid, ms))
            results["ms_stage%d_nu%d_grid%d_r%d" % (stage, nu, grid, rnd)] = ms
s.step_device()
for nm in (("spmv_St_r", "spmv_St") if which == "St" else ("spmv_S",)):
    results["product_seq_" + nm] = s.bench_kernel("seq:" + nm, 20)[0]
    results["product_replayed_" + nm] = s.bench_kernel(nm, 20)[0]
log(json.dumps(results))
counts["results"] = results
json.dump(counts, open(os.path.join(ROOT, "gpurun_out", tag + "_counts.json"), "w"), indent=1)
s.close()
