#!/bin/bash
# usage: scripts/pmc_run.sh <tag> "<counters pass 1>" "<counters pass 2>" ...   (environment selects the variant)
# Each pass: rocprofv3 --pmc <counters> on one setup + kernel micro-benchmarks of the 256^3 cavity.  Summary -> gpurun_out/<tag>.json
set -e
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  EXP_STEP=0 rocprofv3 --pmc $C --kernel-trace -d $OUT/p$i -o run --output-format csv -- python3 $R/scripts/exp_layout.py child > $OUT/p$i.log 2>&1
  echo "pass $i done" >> $OUT/progress
done
cd $R
python3 scripts/pmc_table.py $OUT > gpurun_out/$TAG.json
cat gpurun_out/$TAG.json
rm -rf $OUT/p*/  # raw csvs are large
