#!/bin/bash
# exp_stream_instr.sh: TIMING experiment (WRONG RESULTS) on scratch copies of the kernel header — what would the two-unit S kernel gain if
# the per-row streams came in fewer vector memory instructions (one interleaved record per lane instead of columns + codes + McInv code)?
#   lib_nomcc.so : the McInv code byte is not loaded (a lane constant instead)
#   lib_nocode.so: the value codes are not loaded (derived from the column words)
#   lib_noboth.so: both
# Run on the GPU box: PS_LIB=polystokes_amd/variants/lib_<v>.so python scripts/kbench.py 256 spmv_S
set -e
cd $(dirname $0)/../polystokes_amd/csrc
mkdir -p ../variants
for f in ps_context ps_grid ps_tiles ps_blocks; do [ -f _build/$f.o ] || make -s _build/$f.o; done
for V in nomcc nocode noboth; do
  D=_build/$V/src; mkdir -p $D; cp *.hpp *.hip $D/
  sed -i 's|"../../include/polystokes.h"|"../../../../../include/polystokes.h"|' $D/ps_common.hpp
  if [ $V != nocode ]; then
    sed -i 's|const int mA = (int)__builtin_amdgcn_raw_buffer_load_b8(rMcc, (int)rowA, 0, NT ? PS_EPI_AUX : 0);|const int mA = (int)(lane \& 3u);|; s|const int mB = (int)__builtin_amdgcn_raw_buffer_load_b8(rMcc, (int)rowB, 0, NT ? PS_EPI_AUX : 0);|const int mB = (int)(lane \& 1u);|' $D/ps_kernels_spmv.hpp
    grep -q "const int mA = (int)(lane & 3u);" $D/ps_kernels_spmv.hpp
  fi
  if [ $V != nomcc ]; then
    python3 - $D/ps_kernels_spmv.hpp <<'PY'
import sys,re
p=sys.argv[1]; s=open(p).read()
i=s.index("__device__ inline EllRegs ellLoad("); j=s.index("return r;", i)
body=s[i:j]
body=re.sub(r"const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64\(rCode,[^;]*;", "const u32x2 v = {q.x & 0x03030303u, q.y & 0x03030303u};", body)
body=re.sub(r"const unsigned v = __builtin_amdgcn_raw_buffer_load_b32\(rCode,[^;]*;", "const unsigned v = (unsigned)(q.x & 0x03030303u);", body)
body=body.replace("const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(rCol, cb, 0, AUX);\n        const unsigned v = (unsigned)(q.x & 0x03030303u);","const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(rCol, cb, 0, AUX);\n        const unsigned v = (unsigned)(q & 0x03030303u);")
s=s[:i]+body+s[j:]
open(p,"w").write(s)
PY
  fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c $D/ps_solve.hip -o _build/$V/ps_solve.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/lib_$V.so _build/ps_context.o _build/ps_grid.o _build/ps_tiles.o _build/ps_blocks.o _build/$V/ps_solve.o
  echo built $V
done
