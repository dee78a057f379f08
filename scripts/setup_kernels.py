"""Per-setup kernel table from a rocprofv3 --kernel-trace --stats csv of scripts/setup_prof.py (three setups): ms per setup, calls per setup, average."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nset = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tot = 0.
out = []
for r in rows:
    t = float(r["TotalDurationNs"]) / 1e6 / nset
    tot += t
    out.append((t, int(r["Calls"]) / nset, float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]))
out.sort(reverse=True)
for t, c, a, n in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%7.3f ms/setup  %5.1f calls  %8.1f us avg  %s" % (t, c, a, n))
print("device total %.2f ms per setup" % tot)
