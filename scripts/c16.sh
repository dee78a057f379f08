#!/bin/bash
set -e
python3 scripts/kbench.py 256 spmv_S,spmv_St,apply,cg_update_r,cg_update_xp,cg_update_xr,cg_update_p > gpurun_out/c16.log 2>&1
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu >> gpurun_out/c16.log 2>&1
python3 bench.py --no-cpu-baseline > gpurun_out/bench_dx.json 2>> gpurun_out/c16.log
