set -e
R=$PWD
PS_DIST_OVERLAP=0 python3 scripts/cut_overhead.py cavity 256 2x2x2 > gpurun_out/r06_cut_overhead_noovl.log 2>&1
cat gpurun_out/r06_cut_overhead_noovl.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_cut_prof -o run --output-format csv -- python3 $R/scripts/cut_overhead.py cavity 256 2x2x2 > $R/gpurun_out/r06_cut_prof.log 2>&1
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_cut_prof/run_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms %.1f"%(tot/1e6))
for r in rows[:24]: print("%8.1f ms %7s calls %8.1f us avg  %s"%(float(r['TotalDurationNs'])/1e6, r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:90]))
PY
