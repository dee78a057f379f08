#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: a copy of the library with other compile-time switches (A/B runs) -> polystokes_amd/variants/lib_<name>.so
set -e
N=$1; shift
cd $(dirname $0)/../polystokes_amd/csrc
mkdir -p ../variants _build/$N
for f in ps_context ps_grid ps_tiles ps_blocks; do [ -f _build/$f.o ] || make -s _build/$f.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c ps_solve.hip -o _build/$N/ps_solve.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/lib_$N.so _build/ps_context.o _build/ps_grid.o _build/ps_tiles.o _build/ps_blocks.o _build/$N/ps_solve.o
