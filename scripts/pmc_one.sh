#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of one bench step under the given environment (two rocprofv3 --pmc passes, no other domains).
#   usage: scripts/pmc_one.sh <tag> "ENV=.. ENV2=.."     -> gpurun_out/<tag>.txt: per kernel launches, 2*FETCH+WRITE bytes per launch
TAG=$1; ENVS=$2
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in $ENVS; do export $S; done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --maxit 20 --no-cpu-baseline --no-strong-512 > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o run --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --maxit 20 --no-cpu-baseline --no-strong-512 > $OUT/w.log 2>&1
cd $R
python3 - "$OUT" <<'PY' > gpurun_out/$TAG.txt
import csv, glob, sys, collections
out = sys.argv[1]
def coll(sub, ctr):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == ctr:
                n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                acc[n][0] += 1; acc[n][1] += float(r["Counter_Value"])
    return acc
f, w = coll("pmc_fetch", "FETCH_SIZE"), coll("pmc_write", "WRITE_SIZE")
for n in sorted(f, key=lambda k: -f[k][1]):
    if f[n][0] >= 10 and n in w:
        print("%-44s launches %4d  traffic %.3f GB per launch (2*FETCH %.3f + WRITE %.3f)" % (n[:44], f[n][0], (2 * f[n][1] / f[n][0] + w[n][1] / w[n][0]) * 1024 / 1e9, 2 * f[n][1] / f[n][0] * 1024 / 1e9, w[n][1] / w[n][0] * 1024 / 1e9))
PY
