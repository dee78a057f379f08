#!/bin/bash
# FETCH_SIZE of the pipelined SpMV kernels with the plain and the XCD-aware chunk walk
set -e
R=$PWD
./scripts/_bin/t_xcc > gpurun_out/xcc.log 2>&1
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  export PS_XCD=$x
  python3 $R/scripts/kbench.py 256 spmv_S,spmv_St > $R/gpurun_out/xcd_time$x.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_xcd$x -o run --output-format csv -- python3 $R/scripts/kbench.py 256 spmv_S,spmv_St > $R/gpurun_out/pmc_xcd$x.log 2>&1
done
