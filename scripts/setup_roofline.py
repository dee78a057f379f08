#!/usr/bin/env python3
"""Setup-stage kernels of one 256^3 cavity step against their rooflines.

usage (GPU box):  rocprofv3 --kernel-trace --stats -d DIR -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
                  python3 scripts/setup_roofline.py DIR/run_kernel_stats.csv [res] > profiles/r03_setup_roofline.json

Durations: rocprofv3's per-kernel averages of that run (setup runs twice in it: bench.py's device step and its host-boundary step).
Algorithmic bytes / flops per launch: the models below (every input array element and every output element once; neighbour
reads that fall on elements another thread of the launch reads anyway are not counted twice), evaluated with the dimensions of the
same scene set up here.  Peak: 8 TB/s HBM, 78.6 TFLOP/s fp64 vector / matrix (MI355X_MICROARCH.md)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import polystokes_amd
from polystokes_amd import _abi as abi, scenes

stats_csv = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
s.setup()
N = n ** 3                                   # cells
F = (n + 1) * n * n                          # faces of one axis
E = (n + 1) * (n + 1) * n                    # edges of one axis
nA, nP, nT, R = s.nA, s.nP, s.nT, s.nRegions
nsys = nP + nT
ptrS, ptrT = s.array("S.ptr"), s.array("St.ptr")
rowsS, nnz = int(ptrS.size - 1), int(ptrS[-1])
nskin = rowsS - nA
nE = [int((s.array(k + "ActiveIndices") >= 0).sum()) for k in ("edgeYZ", "edgeXZ", "edgeXY")]
lenT = np.diff(ptrT)
runs = s.array("streamRuns")
chunksS, chunksT = (rowsS + 255) // 256, (nsys + 255) // 256
red_faces = int(3 * R * 14 * 14 * 15) if R else 0           # faces of a 14^3 tile (two of them shared with nobody: tiles are separated by the padding)
box_faces = int(3 * R * 15 * 14 * 14) if R else 0           # positions of the three face boxes a tile's work items walk
s.close()

HBM, FP64 = 8.0e12, 78.6e12
K = {}   # name prefix -> (launches per setup, bytes per launch, flops per launch, model text)


def add(name, launches, nbytes, flops, model):
    K[name] = (launches, float(nbytes), float(flops), model)


add("k_sdf_weights7", 2, 4 * N + 4 * (N + 3 * F + 3 * E), 0, "SDF once (4 N) + one weight per sample point of all seven grids (4 B each)")
add("k_cc_step", 12, N * (1 + 4 + 4), 0, "link byte, component label in / out: 9 B per cell")
add("k_S_count", 3, F * 36 + rowsS / 3.0 * 4, 0, "faceRow, face weight, 2 cells x (lw, sys) and 2 x 2 edges x (lab, lw) amortised: 36 B per face; one count per row")
add("k_S_fill", 3, F * 52 + nnz / 3.0 * 13 + rowsS / 3.0 * 28, 0, "inputs 52 B per face (faceRow, fw, vel, cells: lw lab sys sysT, edges: lab lw sys); per entry col 4 + val 8 + code 1; per row McInv, rhs, old u, ptr")
add("k_St_cells<true>", 1, N * 52 + float(lenT.sum()) * 13 * (4.0 * nP / max(nsys, 1)) + 4 * nP * 20, 0, "per cell 52 B of inputs; 13 B per entry of the four cell rows; rhs, uInv, ptr per row")
add("k_St_cells<false>", 1, N * 40 + 4 * nP * 4, 0, "inputs without values; one count per row")
add("k_St_edges<true>", 3, E * 32 + nnz * (float(sum(nE)) / max(nsys, 1)) / 3.0 * 13 + sum(nE) / 3.0 * 20, 0, "per edge 32 B of inputs; 13 B per entry; rhs, uInv, ptr per row")
add("k_St_edges<false>", 3, E * 24 + sum(nE) / 3.0 * 4, 0, "inputs; one count per row")
add("k_col16_build", 2, 8 * nnz + 5 * (rowsS + nsys) / 2.0 + 80 * (chunksS + chunksT) / 2.0, 0, "ptr, col (4), code (1) in; col16 (2), code4 (1), len8, 80 B per chunk out — mean of S and St")
add("k_chunk_hash", 2, 3 * nnz + (rowsS + nsys) / 2.0, 0, "the payload once: 3 B per entry + 1 B per row")
add("k_chunk_share", 2, 2 * (3 * nnz + (rowsS + nsys) / 2.0), 0, "own payload + the representative's (mostly from cache)")
add("k_ell_fill", 2, 3 * 1.3e6 * 2, 0, "only the distinct runs are laid out (about 1.3 M entries): negligible")
add("k_il_assign", 2, (4096 * 7 + 4096 * 3) / 2.0 * ((n + 16) // 16) ** 3 * 4 * 2, 0, "one label read + one index written per lattice position and group (7 DOF kinds / 3 face axes)")
add("k_il_count", 2, (4096 * 7 + 4096 * 3) / 2.0 * ((n + 16) // 16) ** 3 * 4, 0, "one label read per lattice position and group")
add("k_jacobi_diag", 1, nnz * 12 + nsys * 16 + float(nskin) * 8, 2.0 * 26 * 26 * float(nnz) * (float(nskin) / max(rowsS, 1)) * 0.28, "St rows (12 B per entry) + diag out; flops: a 26 x 26 quadratic form per entry on a skin row (about 28 % of the entries of skin rows' columns)")
add("k_region_outer_mfma<0>", 1, box_faces * 16 + R * 3 * 703 * 8, 2.0 * 32 * 32 * box_faces, "labels / region ids per box position; MFMA flops 2 x 32 x 32 per staged position (padded 26 -> 32)")
add("k_region_outer_mfma<1>", 1, box_faces * 16 + R * 3 * 703 * 8, 2.0 * 32 * 32 * box_faces, "as <0>")
add("k_region_outer_mfma<2>", 1, box_faces * 120 + R * 3 * 703 * 8, 2.0 * 32 * 32 * box_faces + red_faces * 20 * 90.0, "per reduced face up to 20 basis rows of neighbours (about 90 flops each) + labels, region ids, viscosity samples (about 120 B); MFMA flops as <0>")
add("k_skin<true>", 1, R * 4096 * 3 * 40, 0, "three candidate faces per box position, 40 B of inputs each (evaluated twice: classify + number)")
add("k_skin<false>", 1, R * 4096 * 3 * 40, 0, "as <true>, counted once")

rows = list(csv.DictReader(open(stats_csv)))
out = {"scene": "cavity %d^3, tile 16 / pad 2" % n, "peaks": {"hbm_bytes_per_s": HBM, "fp64_flops_per_s": FP64},
       "dims": {"cells": N, "faces_per_axis": F, "active_faces": nA, "skin_rows": nskin, "nnz": nnz, "dofs": nsys, "regions": R},
       "source": os.path.basename(os.path.dirname(stats_csv)) + "/" + os.path.basename(stats_csv), "kernels": {}}
total_ms = 0.0
for name, (launches, nbytes, flops, model) in K.items():
    hit = [r for r in rows if name.split("<")[0] in r["Name"] and (("<" not in name) or name[name.index("<"):] in r["Name"])]
    if not hit:
        continue
    avg_ns = sum(float(r["AverageNs"]) * int(r["Calls"]) for r in hit) / sum(int(r["Calls"]) for r in hit)
    t = avg_ns * 1e-9
    ent = {"launches_per_setup": launches, "avg_ms": avg_ns * 1e-6, "ms_per_setup": avg_ns * 1e-6 * launches, "algorithmic_bytes": nbytes,
           "GBps": nbytes / t / 1e9, "frac_hbm": nbytes / t / HBM, "model": model}
    if flops:
        ent.update({"flops": flops, "TFLOPs": flops / t / 1e12, "frac_fp64": flops / t / FP64, "bound": "fp64" if flops / FP64 > nbytes / HBM else "hbm"})
    else:
        ent["bound"] = "hbm"
    total_ms += ent["ms_per_setup"]
    out["kernels"][name] = ent
out["listed_ms_per_setup"] = total_ms
print(json.dumps(out, indent=1))
