#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats, and the FETCH_SIZE / WRITE_SIZE counter passes
# (separate --pmc runs, no other trace domains).  Results land in gpurun_out/<tag>/ ; copy the summaries to profiles/.
# The kernel stats are those of a WARM process (1 warm-up + 3 timed steps: VERDICT r04 weak #4 — the single profiled step of r04 read 6 % faster than
# the bench steps of the same call; profiles/r05_cold_warm.md shows a process repeats to 0.1 % from its second step on).
#   usage: scripts/profile_round.sh <tag>
set -e
TAG=${1:-prof}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" > $OUT/progress
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-512 --no-other-preconditioners > $OUT/stats.log 2>&1
echo "stats done" >> $OUT/progress
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --maxit 20 --no-cpu-baseline --no-strong-512 --no-other-preconditioners > $OUT/pmc_fetch.log 2>&1
echo "fetch done" >> $OUT/progress
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --maxit 20 --no-cpu-baseline --no-strong-512 --no-other-preconditioners > $OUT/pmc_write.log 2>&1
echo "write done" >> $OUT/progress
cd $R
python3 scripts/pmc_summarize.py $OUT > $OUT/pmc_traffic.json
