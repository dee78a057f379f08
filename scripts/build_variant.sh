#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: a copy of the library with other compile-time switches (A/B runs) -> polystokes_amd/variants/lib_<name>.so
# -DPS_EXP_GATHER_MASK=<mask> (timing experiment, WRONG RESULTS: every gather of the SpMV kernels falls into the first mask + 8 bytes of its
# vector — what a perfect gather locality could buy at most) is applied to a SCRATCH COPY of the kernel header: the product sources do not carry it.
set -e
N=$1; shift
cd $(dirname $0)/../polystokes_amd/csrc
mkdir -p ../variants _build/$N
for f in ps_context ps_grid ps_tiles ps_blocks; do [ -f _build/$f.o ] || make -s _build/$f.o; done
SRC=ps_solve.hip
INC=""
if echo "$@" | grep -q PS_EXP_GATHER_MASK; then
  mkdir -p _build/$N/src
  cp *.hpp *.hip _build/$N/src/
  sed -i 's|^__device__ inline double bufGatherF64(__amdgpu_buffer_rsrc_t r, unsigned byteOff) {$|&\n    byteOff \&= PS_EXP_GATHER_MASK;|' _build/$N/src/ps_kernels_spmv.hpp
  grep -q "byteOff &= PS_EXP_GATHER_MASK" _build/$N/src/ps_kernels_spmv.hpp
  sed -i 's|"../../include/polystokes.h"|"../../../../../include/polystokes.h"|' _build/$N/src/ps_common.hpp
  SRC=_build/$N/src/ps_solve.hip
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c $SRC -o _build/$N/ps_solve.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/lib_$N.so _build/ps_context.o _build/ps_grid.o _build/ps_tiles.o _build/ps_blocks.o _build/$N/ps_solve.o
