# rocprofv3 kernel table of the in-process proxy (scripts/cut_overhead.py) for one decomposition; env passes through (PS_DIST_OVERLAP=1: split launches)
# usage: prof_cut2.sh <tag> <scene> <res> <dims>
set -e
R=$PWD
TAG=$1; SCENE=${2:-cavity}; RES=${3:-256}; DIMS=${4:-2x2x2}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o run --output-format csv -- python3 $R/scripts/cut_overhead.py $SCENE $RES $DIMS > $R/gpurun_out/$TAG.log 2>&1
cd $R
cat gpurun_out/$TAG.log
python3 - $TAG <<'PY'
import csv, sys
tag = sys.argv[1]
rows=list(csv.DictReader(open('gpurun_out/%s/run_kernel_stats.csv' % tag)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms %.1f"%(tot/1e6))
for r in rows[:16]: print("%8.1f ms %7s calls %8.1f us avg  %s"%(float(r['TotalDurationNs'])/1e6, r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
