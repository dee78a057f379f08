#!/bin/bash
# FETCH_SIZE and time of the persistent SpMV kernels as a function of the XCD run length G
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for x in 0 16 128 1024 8192; do
  export PS_XCD=$x
  python3 $R/scripts/kbench.py 256 spmv_S,spmv_St > $R/gpurun_out/xcd_time$x.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_xcd$x -o run --output-format csv -- python3 $R/scripts/kbench.py 256 spmv_S,spmv_St > $R/gpurun_out/pmc_xcd$x.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, collections, glob
for x in (0, 16, 128, 1024, 8192):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(f"gpurun_out/pmc_xcd{x}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "spmv" in r["Kernel_Name"] and "pipe" in r["Kernel_Name"]:
                k = "St" if "St_pipe" in r["Kernel_Name"] else "S"
                agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    t = open(f"gpurun_out/xcd_time{x}.log").read().strip().splitlines()[-2:]
    print("G", x, {k: round(2 * v[1] / v[0] * 1024 / 1e9, 3) for k, v in agg.items()}, "GB fetched (x2) |", " ; ".join(t))
PY
