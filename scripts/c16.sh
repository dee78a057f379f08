#!/bin/bash
set -e
python3 scripts/kbench.py 256 apply,cg_update_r,cg_update_xp > gpurun_out/c16.log 2>&1
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu >> gpurun_out/c16.log 2>&1
for n in 32 64 128 256; do python3 bench.py --res $n --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print($n, round(d['value'],2), d['cg_iterations'], round(d['cg_iters_per_s']), round(d['stage_ms']['solve'],2))" >> gpurun_out/c16.log; done
