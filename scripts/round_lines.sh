#!/bin/bash
# Bench lines of the other BASELINE configurations and of the Chebyshev option (one GPU), with kernel stats for the small one.
#   usage: scripts/round_lines.sh <tag>       -> gpurun_out/<tag>/bench_{coil128,coil256,spheres256,chebyshev256}.json, stats_coil128/, stats_chebyshev/
TAG=${1:-lines}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --scene coil --res 128 --no-cpu-baseline --no-strong-512 --steps 5 > $OUT/bench_coil128.json 2> $OUT/bench_coil128.err; echo "coil128 $?" >> $OUT/progress
python3 bench.py --scene coil --res 256 --no-cpu-baseline --no-strong-512 --steps 5 > $OUT/bench_coil256.json 2> $OUT/bench_coil256.err; echo "coil256 $?" >> $OUT/progress
python3 bench.py --scene spheres --res 256 --no-cpu-baseline --no-strong-512 --steps 5 > $OUT/bench_spheres256.json 2> $OUT/bench_spheres256.err; echo "spheres256 $?" >> $OUT/progress
python3 bench.py --precond chebyshev --no-cpu-baseline --no-strong-512 --no-other-preconditioners > $OUT/bench_chebyshev256.json 2> $OUT/bench_chebyshev256.err; echo "chebyshev $?" >> $OUT/progress
python3 bench.py --no-cpu-baseline --no-strong-512 --no-other-preconditioners > $OUT/bench_jacobi256_same_box.json 2> $OUT/bench_jacobi256.err; echo "jacobi $?" >> $OUT/progress
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats_coil128 -o run --output-format csv -- python3 $R/bench.py --scene coil --res 128 --steps 1 --warmup 0 --no-cpu-baseline --no-strong-512 > $OUT/stats_coil128.log 2>&1; echo "stats coil128 $?" >> $OUT/progress
rocprofv3 --kernel-trace --stats -d $OUT/stats_chebyshev -o run --output-format csv -- python3 $R/bench.py --precond chebyshev --steps 1 --warmup 0 --no-cpu-baseline --no-strong-512 --no-other-preconditioners > $OUT/stats_chebyshev.log 2>&1; echo "stats cheb $?" >> $OUT/progress
