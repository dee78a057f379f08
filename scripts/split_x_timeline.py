"""Timeline of a few PCG iterations from a rocprofv3 kernel-trace csv (PS_SPLIT_X runs: does the x update overlap the next S?).
usage: split_x_timeline.py <kernel_trace.csv> [first iteration] [iterations]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 3
short = lambda n: n.split("(")[0].split("<")[0].replace("void ", "")[:28]
# find the S launches of the LAST solve
S = [i for i, r in enumerate(rows) if "k_spmv_S_ell" in r["Kernel_Name"]]
S = S[len(S) // 2:]
i0, i1 = S[first], S[first + cnt]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f -> %9.1f us  (%7.1f)  queue %-3s %s" % (a, b, b - a, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
