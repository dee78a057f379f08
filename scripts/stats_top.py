"""Top kernels of a rocprofv3 kernel stats csv: name, calls, average us."""
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 5]:
    print("%-64s calls %6s avg %9.2f us" % (r["Name"].replace("(anonymous namespace)::", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
