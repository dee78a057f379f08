#!/usr/bin/env python3
"""Ground truth for the St-kernel's t traffic: distinct 128-B lines of t touched per group of 256-DOF chunks, split into the
active-row part and the skin-row part of t (internal numbering), cavity N^3."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import _abi as abi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sc, p = scenes.cavity(n, tile=16, pad=2, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0); s.upload(sc, p); s.setup()
ptr, col = s.array("St.ptr").astype(np.int64), s.array("St.col").astype(np.int64)
nA = s.nA; rows = len(ptr) - 1; ncols = int(col.max()) + 1
print("DOF rows", rows, "face rows", ncols, "nA", nA, "nnz", len(col))
line = col // 16
rowof = np.repeat(np.arange(rows), np.diff(ptr))
act = col < nA
tl = ncols / 16
for G in (1, 4, 16, 64):
    for nm, m in (("active", act), ("skin", ~act), ("all", np.ones_like(act))):
        key = (rowof[m] // (256 * G)) * (ncols // 16 + 2) + line[m]
        print("G=%3d %-6s sum of distinct t lines per group = %.2fx of that part's lines (%.2fx of all t lines)" % (
            G, nm, len(np.unique(key)) / max(len(np.unique(line[m])), 1), len(np.unique(key)) / tl))
