#!/bin/bash
set -e
for g in 0 4 8 16 32 64 128 512; do
  export PS_XCD=$g
  echo "G=$g" >> gpurun_out/xcd_sweep.log
  python3 scripts/kbench.py 256 spmv_S,spmv_St,apply >> gpurun_out/xcd_sweep.log 2>&1
done
PS_XCD=32 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cavity or blob" >> gpurun_out/xcd_sweep.log 2>&1
