#!/bin/bash
set -e
for g in 0 8 16 32 64 128; do
  export PS_XCD=$g
  echo "G=$g" >> gpurun_out/xcd_sweep.log
  python3 scripts/kbench.py 256 spmv_S,spmv_St,apply >> gpurun_out/xcd_sweep.log 2>&1
done
