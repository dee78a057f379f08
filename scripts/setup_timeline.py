#!/usr/bin/env python3
"""Timeline of the LAST setup of a rocprofv3 --kernel-trace [--memory-copy-trace] run of scripts/setup_prof.py: where the wall time
of setup goes that is not kernel time.  usage: setup_timeline.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [min gap us]"""
import csv, glob, sys
d = sys.argv[1]
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ev = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]))
for fn in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "")))
ev.sort()
starts = [i for i, e in enumerate(ev) if "k_sdf_weights" in e[2] and (i == 0 or "k_sdf_weights" not in ev[i - 1][2])]
a = starts[-1]
ev = ev[a:]
t0 = ev[0][0]
busy = 0
prev_end = t0
gaps = []
for s, e, n in ev:
    if s - prev_end > mingap * 1e3:
        gaps.append(((prev_end - t0) / 1e6, (s - prev_end) / 1e3, n))
    busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
wall = (prev_end - t0) / 1e6
print("last setup: %d events, wall %.2f ms, device busy %.2f ms, idle %.2f ms" % (len(ev), wall, busy / 1e6, wall - busy / 1e6))
print("gaps > %.0f us:  at ms | gap us | next event" % mingap)
for at, g, n in gaps:
    print("  %8.2f  %8.1f  %s" % (at, g, n[:90]))
import collections
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    tot[n][0] += 1; tot[n][1] += e - s
print("per kernel (last setup):  ms | launches | name")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %7.3f  %4d  %s" % (t / 1e6, c, n[:100]))
