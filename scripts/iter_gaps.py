#!/usr/bin/env python3
"""Idle time between the kernels of the PCG iteration in a rocprofv3 kernel trace: for every pair (kernel, next kernel) of the solve loop the
mean / median gap from the end of one to the start of the next, and the share of the loop's wall time the gaps take.
usage: iter_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, statistics, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    for k in ("k_spmv_S_ell2", "k_spmv_St_ell2", "k_tile_apply", "k_cg_update_xp_u", "k_cg_check"):
        if k in n: return k
    return None
loop = [(s, e, short(n)) for s, e, n in rows if short(n)]
gaps, busy = {}, 0
for (s0, e0, n0), (s1, e1, n1) in zip(loop, loop[1:]):
    busy += e0 - s0
    if n1 == "k_spmv_S_ell2" and n0 == "k_cg_check": continue      # the host round trip of a batch end: listed apart
    gaps.setdefault((n0, n1), []).append(s1 - e0)
wall = loop[-1][1] - loop[0][0]
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 5: continue
    tot += sum(v)
    print("%-18s -> %-18s n %5d  mean %7.2f us  median %7.2f us  max %8.1f us" % (k[0], k[1], len(v), statistics.mean(v) / 1e3, statistics.median(v) / 1e3, max(v) / 1e3))
batch = [s1 - e0 for (s0, e0, n0), (s1, e1, n1) in zip(loop, loop[1:]) if n0 == "k_cg_check" and n1 == "k_spmv_S_ell2"]
if batch: print("batch ends (k_cg_check -> next S, host round trip): n %d  mean %.1f us" % (len(batch), statistics.mean(batch) / 1e3))
print("loop wall %.2f ms, kernels busy %.2f ms, gaps inside iterations %.2f ms (%.2f %%), batch ends %.2f ms" % (wall / 1e6, busy / 1e6, tot / 1e6, 100. * tot / wall, sum(batch) / 1e6))
