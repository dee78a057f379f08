#!/bin/bash
# Same-box A/B of library builds under rocprofv3 kernel stats: one bench step per build (PS_LIB), the per-kernel averages side by side.
#   usage: scripts/ab_stats.sh <tag> <lib1.so> <lib2.so> ...      (paths relative to the repo root; "HEAD" = the in-tree build)
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for L in "$@"; do
  N=$(basename $L .so)_$round
  if [ "$L" = "HEAD" ]; then unset PS_LIB; else export PS_LIB=$R/$L; fi
  rocprofv3 --kernel-trace --stats -d $OUT/$N -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-strong-512 > $OUT/$N.json 2> $OUT/$N.err
  echo "$N done" >> $OUT/progress
done
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print(os.path.basename(d.rstrip("/")))
    for r in rows[:6]:
        print("   %-70s calls %6s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
