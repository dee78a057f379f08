#!/bin/bash
set -e
for g in 0 -1 -2 -3 -4; do
  export PS_XCD=$g
  echo "G=$g" >> gpurun_out/fake.log
  python3 scripts/kbench.py 256 spmv_St >> gpurun_out/fake.log 2>&1
done
