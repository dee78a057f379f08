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
import polystokes_amd
from polystokes_amd import scenes
T0 = time.time()


def log(*a):
    print("[%6.1f s]" % (time.time() - T0), *a, flush=True)


sc, p = getattr(scenes, scene)(res, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
nx, ny, nz = sc.nx, sc.ny, sc.nz
nA, nP = s.nA, s.nP
ptr = s.array("St.ptr").astype(np.int64)
col = s.array("St.col").astype(np.int64)
code = s.array("St.code").astype(np.int8)
rows = ptr.size - 1
nnz = col.size
tLen = int(col.max()) + 1
log("St: rows %d nnz %d cols %d (nA %d)" % (rows, nnz, tLen, nA))
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
dofBlk = block_of(["centerActiveIndices"] * 4 + ["edgeYZActiveIndices", "edgeXZActiveIndices", "edgeXYActiveIndices"], [cd] * 4 + ed, sysPerm,
                  [0, nP, 2 * nP, 3 * nP, 4 * nP, 4 * nP + nE[0], 4 * nP + nE[0] + nE[1]], rows)
assert (faceBlk >= 0).all() and (dofBlk >= 0).all()
assert (np.diff(dofBlk) >= 0).all() and (np.diff(faceBlk) >= 0).all(), "the internal numbering is not block-major"
# the lattice blocks that hold DOF rows
bstart = np.concatenate(([0], np.flatnonzero(np.diff(dofBlk)) + 1))
bid = dofBlk[bstart]
nB = bstart.size
d0 = bstart
d1 = np.concatenate((bstart[1:], [rows]))
# active face rows of each of those blocks: [a0, a1)
a0 = np.searchsorted(faceBlk, bid, "left")
a1 = np.searchsorted(faceBlk, bid, "right")
reg = s.array("reducedRowRegion").astype(np.int64)
assert (np.diff(reg) >= 0).all()
nReg = int(reg.max()) + 1 if reg.size else 0
regPtr = np.searchsorted(reg, np.arange(nReg + 1), "left")
lens = np.diff(ptr)
rowB = np.repeat(np.arange(nB), d1 - d0)                  # block (dense id) of every DOF row
nzB = np.repeat(rowB, lens)                                # ... of every entry
# the tile of a block = the region its rows reference most
skin = col >= nA
key = nzB[skin] * (nReg + 1) + reg[col[skin] - nA]
uk, cnt = np.unique(key, return_counts=True)
ub, ur = uk // (nReg + 1), uk % (nReg + 1)
best = np.full(nB, -1, np.int64)
order = np.lexsort((cnt, ub))                              # last entry per block = most frequent
lastOf = np.flatnonzero(np.concatenate((ub[order][1:] != ub[order][:-1], [True])))
best[ub[order][lastOf]] = ur[order][lastOf]
s0 = np.where(best >= 0, nA + regPtr[np.maximum(best, 0)], 0)
s1 = np.where(best >= 0, nA + regPtr[np.maximum(best, 0) + 1], 0)
# ranges start on even entries (16-byte pieces), lengths padded to whole 128-double pieces
a0e = a0 & ~1
lenA = ((a1 - a0e + 127) // 128) * 128
lenA[a1 == a0] = 0
s0e = s0 & ~1
lenS = ((s1 - s0e + 127) // 128) * 128
lenS[s1 == s0] = 0
inA = (col >= a0e[nzB]) & (col < (a0e + lenA)[nzB])
inS = (~inA) & (col >= s0e[nzB]) & (col < (s0e + lenS)[nzB])
rest = ~(inA | inS)
hkey = (nzB[rest] << 32) | (col[rest] >> 1)
hu, hinv = np.unique(hkey, return_inverse=True)
hb = hu >> 32
haloBase = np.searchsorted(hb, np.arange(nB), "left")
nPairs = np.searchsorted(hb, np.arange(nB), "right") - haloBase
haloPairs = (hu & 0xffffffff).astype(np.int32)
pos = np.empty(nnz, np.int64)
pos[inA] = 2 + col[inA] - a0e[nzB[inA]]
pos[inS] = 2 + lenA[nzB[inS]] + col[inS] - s0e[nzB[inS]]
b_r = nzB[rest]
pos[rest] = 2 + lenA[b_r] + lenS[b_r] + 2 * (hinv - haloBase[b_r]) + (col[rest] & 1)
img = 2 + lenA + lenS + ((nPairs + 63) // 64) * 128
log("blocks %d; image doubles per block: mean %.0f p99 %.0f max %d; halo pairs mean %.0f max %d; in A %.3f in S %.3f halo %.3f of the entries"
    % (nB, img.mean(), np.percentile(img, 99), img.max(), nPairs.mean(), nPairs.max(), inA.mean(), inS.mean(), rest.mean()))
assert pos.max() < 65536
counts = {"scene": "%s%d" % (scene, res), "blocks": int(nB), "rows": int(rows), "nnz": int(nnz), "image_doubles_mean": float(img.mean()), "image_doubles_p99": float(np.percentile(img, 99)),
          "image_doubles_max": int(img.max()), "halo_pairs_mean": float(nPairs.mean()), "halo_pairs_max": int(nPairs.max()),
          "entries_from_active_range": float(inA.mean()), "entries_from_skin_range": float(inS.mean()), "entries_from_halo": float(rest.mean()),
          "staged_doubles_per_gathered_entry": float(img.sum() / nnz), "staged_doubles_over_t_entries": float(img.sum() / tLen),
          "dma_pieces_per_block_mean": float((img / 128).mean())}
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
import torch
dev = "cuda"
gen = np.random.default_rng(7)
tvec = np.zeros(tLen + 256); tvec[:tLen] = gen.standard_normal(tLen)
pvec = gen.standard_normal(rows); rvec = gen.standard_normal(rows)
ucode = gen.integers(0, 256, rows).astype(np.uint8); udict = gen.random(256)
dinv = gen.random(rows).astype(np.float32)
scale, alpha = 0.5, 0.37
import scipy.sparse as sp
St = sp.csr_matrix((code.astype(np.float64) * scale, col, ptr), shape=(rows, tLen))
y = -(St @ tvec[:tLen]) - 0.5 * udict[ucode] * pvec
rexp = rvec - alpha * y
L = C.CDLL(so)
L.st_lds_launch.restype = C.c_int
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
T = dict(desc=tt(desc), units=tt(unitsC), ecol=tt(ecolC.view(np.int16)), ecode=tt(ecodeC), halo=tt(haloPairs if haloPairs.size else np.zeros(1, np.int32)), t=tt(tvec), p=tt(pvec),
         uc=tt(ucode), ud=tt(udict), dinv=tt(dinv))
ldsBytes = int(((img.max() * 8 + 1023) // 1024) * 1024)
assert ldsBytes <= 80 * 1024 - 4096, "image of %d bytes: more than two workgroups per CU can hold" % ldsBytes
results = {"lds_bytes": ldsBytes}


def launch(stage, grid, reps_):
    r = tt(rvec)
    part = torch.zeros(2 * grid, dtype=torch.float64, device=dev)
    ms = C.c_float(0)
    rc = L.st_lds_launch(C.c_int(stage), C.c_int(grid), C.c_int(ldsBytes), C.c_void_p(T["desc"].data_ptr()), C.c_int(nB), C.c_void_p(T["units"].data_ptr()),
                         C.c_void_p(T["ecol"].data_ptr()), C.c_void_p(T["ecode"].data_ptr()), C.c_uint(ecolC.nbytes), C.c_uint(ecodeC.nbytes), C.c_void_p(T["halo"].data_ptr()),
                         C.c_void_p(T["t"].data_ptr()), C.c_int(tLen), C.c_void_p(T["p"].data_ptr()), C.c_void_p(T["uc"].data_ptr()), C.c_void_p(T["ud"].data_ptr()),
                         C.c_void_p(r.data_ptr()), C.c_void_p(T["dinv"].data_ptr()), C.c_int(rows), C.c_double(scale), C.c_double(alpha), C.c_void_p(part.data_ptr()),
                         C.c_int(reps_), C.byref(ms))
    assert rc == 0, rc
    return r.cpu().numpy(), part.cpu().numpy(), ms.value


for stage in (0, 1):
    rgot, part, _ = launch(stage, 512, 1)
    err = np.abs(rgot - rexp).max() / np.abs(rexp).max()
    rr = part[:512].sum()
    log("stage %d: max rel err of r %.2e; r.r %.12e vs %.12e" % (stage, err, rr, (rexp * rexp).sum()))
    assert err < 1e-12
    results["err_stage%d" % stage] = float(err)
for stage in (0, 1):
    for grid in (256, 512, 768, 1024):
        _, _, ms = launch(stage, grid, 20)
        log("stage %d grid %4d: %.4f ms" % (stage, grid, ms))
        results["ms_stage%d_grid%d" % (stage, grid)] = ms
s.step_device()
for nm in ("spmv_St_r", "spmv_St"):
    results["product_seq_" + nm] = s.bench_kernel("seq:" + nm, 20)[0]
    results["product_replayed_" + nm] = s.bench_kernel(nm, 20)[0]
log(json.dumps(results))
counts["results"] = results
json.dump(counts, open(os.path.join(ROOT, "gpurun_out", tag + "_counts.json"), "w"), indent=1)
s.close()
