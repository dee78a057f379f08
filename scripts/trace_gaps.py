"""Per-kernel average duration and average gap to the previous kernel from a rocprofv3 kernel trace csv (last solve only)."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = ("k_spmv_S_pipe", "k_tile_apply", "k_spmv_St_pipe", "k_cg_update_r", "k_cg_update_xp")
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in names)]
rows = rows[len(rows) // 2:]          # second solve
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    k = next(k for k in names if k in r["Kernel_Name"])
    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if prev is not None: gap[k].append(int(r["Start_Timestamp"]) - prev)
    prev = int(r["End_Timestamp"])
tot = 0
for k in names:
    if dur[k]:
        d = sum(dur[k]) / len(dur[k]) / 1e3; g = sum(gap[k]) / max(len(gap[k]), 1) / 1e3
        tot += d + g
        print("%-16s %5d launches  avg %.2f us  gap before %.2f us" % (k, len(dur[k]), d, g))
print("sum per iteration %.1f us" % tot)
