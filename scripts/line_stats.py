#!/usr/bin/env python3
"""Ground truth for the S-kernel's x traffic: distinct 128-B lines of x touched per 256-row chunk / per group of chunks,
for the active rows and the skin rows of S (internal numbering), cavity N^3."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polystokes_amd
from polystokes_amd import _abi as abi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sc, p = scenes.cavity(n, tile=16, pad=2, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0); s.upload(sc, p); s.setup()
ptr, col = s.array("S.ptr").astype(np.int64), s.array("S.col").astype(np.int64)
nA = s.nA; rows = len(ptr) - 1; nsys = s.nP + s.nT
print("rows", rows, "nA", nA, "n", nsys, "nnz", len(col), "nnz active", ptr[nA], "x lines", nsys // 16)
line = col // 16
rowof = np.repeat(np.arange(rows), np.diff(ptr))
def distinct_per_group(mask, G):
    grp = rowof[mask] // (256 * G)
    key = grp * (nsys // 16 + 1) + line[mask]
    return len(np.unique(key))
act = rowof < nA
la, ls = np.unique(line[act]), np.unique(line[~act])
print("distinct lines: active %d, skin %d, both %d, union %d" % (len(la), len(ls), len(np.intersect1d(la, ls)), len(np.union1d(la, ls))))
for G in (1, 2, 4, 8, 16, 32, 64, 128):
    print("group of %3d chunks: sum of distinct lines per group: active %.2fx, skin %.2fx of all x lines" % (
        G, distinct_per_group(act, G) / (nsys / 16), distinct_per_group(~act, G) / (nsys / 16)))
# L1-coalescing proxy: distinct lines per (wave-instruction) = per 64 consecutive entries x 4 slots is format specific; report per 64 rows instead
key = (rowof // 64) * (nsys // 16 + 1) + line
print("sum of distinct lines per 64-row group: %.2fx" % (len(np.unique(key)) / (nsys / 16)))
