#!/bin/bash
# Same-box sweep of environment settings: bench.py (3 steps, no CPU baseline) per setting, ms/step + iterations + kernel timings side by side.
#   usage: scripts/env_sweep.sh <tag> "VAR=a VAR2=b" "VAR=c" ...     ("" = defaults; each setting runs twice, interleaved)
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for round in 1 2; do
  i=0
  for S in "$@"; do
    i=$((i+1))
    env $S python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-512 ${BENCH_ARGS} > $OUT/s${i}_$round.json 2> $OUT/s${i}_$round.err
    python3 - "$OUT/s${i}_$round.json" "$S" "$round" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["roofline"]["other_kernels"]
    print("%-40s r%s  %8.1f ms/step  it %4d  solve %8.1f  setup %6.1f | St_r %.4f  S %.4f  xp_u %.4f  tiles %.4f" % (sys.argv[2] or "(default)", sys.argv[3], d["value"], d["cg_iterations"], d["stage_ms"]["solve"],
          d["value"] - d["stage_ms"]["solve"], d["roofline"]["avg_launch_ms"], k["spmv_S"]["ms"], k.get("cg_update_xp_u", {"ms": 0})["ms"], k["tiles"]["ms"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
