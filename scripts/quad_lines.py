#!/usr/bin/env python3
"""L1 tag lookups a gather mapping costs under the quad rule of profiles/r03_spmv_issue.md (one lookup per distinct 128-byte
line among the 4 lanes of an aligned quad, per instruction), computed from the real matrices.
usage: quad_lines.py [scene] [res]      (prints one JSON line)
Mappings: "stream4" = r02 kernels (lane l owns entries 4l..4l+3 of the CSR stream, instruction j takes entry 4l+j);
"row" = one lane per row, instruction k takes entry k of 64 consecutive rows; variants of "row" with other row orders."""
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


def quad_distinct(lines):
    """lines: (Q, 4) int64, negative = inactive lane -> number of distinct non-negative values per quad, summed"""
    a = np.sort(lines, axis=1)
    d = (a[:, 0] >= 0).astype(np.int64)
    for i in range(1, 4):
        d += ((a[:, i] >= 0) & (a[:, i] != a[:, i - 1])).astype(np.int64)
    return int(d.sum())


def ell(ptr, col, order=None, width=8):
    ln = np.diff(ptr)
    rows = ln.size
    M = np.full((rows, width), -1, np.int64)
    for k in range(width):
        m = ln > k
        M[m, k] = col[ptr[:-1][m] + k]
    if order is not None:
        M = M[order]
    pad = (-rows) % 4
    if pad:
        M = np.concatenate([M, np.full((pad, width), -1, np.int64)])
    return M


def row_map(M):
    tot = 0
    for k in range(M.shape[1]):
        tot += quad_distinct((M[:, k] >> 4).reshape(-1, 4) | np.where(M[:, k].reshape(-1, 4) < 0, -1, 0))
    return tot


def stream4(ptr, col):
    nnz = col.size
    pad = (-nnz) % 16
    c = np.concatenate([col.astype(np.int64), np.full(pad, -1, np.int64)])
    c = c.reshape(-1, 4, 4)            # (quad q, lane i, entry j) = entry 16q + 4i + j
    tot = 0
    for j in range(4):
        v = c[:, :, j]
        tot += quad_distinct(np.where(v < 0, -1, v >> 4))
    return tot


out = {"scene": scene, "res": n}
for name, width in (("S", 8), ("St", 6)):
    ptr = s.array(name + ".ptr").astype(np.int64)
    col = s.array(name + ".col").astype(np.int64)
    nnz = int(col.size)
    o = {"nnz": nnz, "rows": int(ptr.size - 1)}
    o["stream4_lookups_per_entry"] = stream4(ptr, col) / nnz
    M = ell(ptr, col, None, width)
    o["row_lookups_per_entry"] = row_map(M) / nnz
    # per-slot breakdown
    o["row_per_slot"] = [quad_distinct(np.where(M[:, k].reshape(-1, 4) < 0, -1, M[:, k].reshape(-1, 4) >> 4)) / max(1, int((M[:, k] >= 0).sum())) for k in range(width)]
    # 64-byte lines instead of 128
    o["row_lookups_per_entry_64B"] = sum(quad_distinct(np.where(M[:, k].reshape(-1, 4) < 0, -1, M[:, k].reshape(-1, 4) >> 3)) for k in range(width)) / nnz
    # adjacent-only merging (a lane merges only with its predecessor in the quad)
    adj = 0
    for k in range(width):
        v = np.where(M[:, k].reshape(-1, 4) < 0, -1, M[:, k].reshape(-1, 4) >> 4)
        d = (v[:, 0] >= 0).astype(np.int64)
        for i in range(1, 4):
            d += ((v[:, i] >= 0) & (v[:, i] != v[:, i - 1])).astype(np.int64)
        adj += int(d.sum())
    o["row_lookups_per_entry_adjacent_only"] = adj / nnz
    # distinct 128-byte lines one 64-row unit touches (the L1 cannot hold much more than one unit's lines: a proxy of the L1 misses)
    rows = M.shape[0]
    padr = (-rows) % 64
    Mu = np.concatenate([M, np.full((padr, width), -1, np.int64)]) if padr else M
    L = np.where(Mu < 0, -1, Mu >> 4).reshape(-1, 64 * width)
    L.sort(axis=1)
    o["unit_distinct_lines_per_entry"] = int(((L[:, 1:] != L[:, :-1]) & (L[:, 1:] >= 0)).sum() + (L[:, 0] >= 0).sum()) / nnz
    # distinct lines of an ITEM = R consecutive 256-row chunks (an LDS line cache staged once per item): lines / entry, largest item
    for R in (1, 2, 4, 8, 16):
        per = 256 * R
        padi = (-rows) % per
        Mi = np.concatenate([M[:rows], np.full((padi, width), -1, np.int64)]) if padi else M[:rows]
        Li = np.where(Mi < 0, -1, Mi >> 4).reshape(-1, per * width)
        Li.sort(axis=1)
        dist = ((Li[:, 1:] != Li[:, :-1]) & (Li[:, 1:] >= 0)).sum(axis=1) + (Li[:, 0] >= 0)
        o["item%d_lines_per_entry" % R] = float(dist.sum() / nnz)
        o["item%d_max_lines" % R] = int(dist.max())
        o["item%d_p99_lines" % R] = int(np.percentile(dist, 99))
    o["vector_lines_per_entry"] = float((int(col.max()) + 16) // 16 / nnz)
    wmax = (Mu >= 0).sum(axis=1).reshape(-1, 64).max(axis=1)
    o["ell_slots_over_nnz"] = float((((wmax + 1) // 2 * 2) * 64).sum() / nnz)
    out[name] = o
print(json.dumps(out))
s.close()
