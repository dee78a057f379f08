#!/bin/bash
# Samples the GPU's clocks / power / temperature once a second while bench.py runs 25 steps: is the 4-6 % spread between the first and
# the later runs on one box a clock effect?     usage: scripts/clock_watch.sh <tag>
TAG=${1:-clock}
OUT=gpurun_out/$TAG; mkdir -p $OUT
( for i in $(seq 1 60); do echo "t=$i $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|memory)" | tr -s ' ' | tr '\n' ';')"; sleep 1; done ) > $OUT/smi.log 2>&1 &
W=$!
python3 bench.py --steps 25 --warmup 0 --no-cpu-baseline --no-strong-512 > $OUT/bench.json 2> $OUT/bench.err
kill $W 2>/dev/null
python3 - "$OUT" <<'PY'
import json, sys, re
d = json.load(open(sys.argv[1] + "/bench.json"))
print("bench: %.1f ms/step over 25 steps" % d["value"])
for l in open(sys.argv[1] + "/smi.log"):
    m = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)|\(avg\)[^:]*: ([0-9.]+)|junction\) \(C\): ([0-9.]+)|memory\) \(C\): ([0-9.]+)", l)
    print(l.strip()[:230])
PY
