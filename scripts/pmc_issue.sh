#!/bin/bash
# Issue-side counter passes on the CG-iteration kernels of the 256^3 cavity: one rocprofv3 --pmc pass per
# counter group (no other trace domain), target = bench.py --maxit 20 (real solve + the in-sequence kernel micro-benchmarks).
#   usage: scripts/pmc_issue.sh <tag> "<group 1>" "<group 2>" ...     (PS_* environment selects the variant)
# Per-kernel averages of every counter -> gpurun_out/<tag>.json (scripts/pmc_table.py); a failed pass is logged and skipped.
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  if rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --maxit 20 --no-cpu-baseline --no-strong-512 > $OUT/p$i.log 2>&1; then
    echo "pass $i ok: $C" >> $OUT/progress
  else
    echo "pass $i FAILED: $C" >> $OUT/progress
  fi
done
cd $R
python3 scripts/pmc_table.py $OUT > gpurun_out/$TAG.json
cat $OUT/progress
rm -rf $OUT/p*/  # raw csvs are large
