#!/bin/bash
# registers of k_tile_apply_w compiled alone (seconds): scripts/tile_regs.sh [-D...]
cd $(dirname $0)/../polystokes_amd/csrc
mkdir -p _build
cat > _build/t_tile.hip <<'EOF'
#include "../ps_context.hpp"
using namespace ps;
namespace {
#include "../ps_kernels_spmv.hpp"
#include "../ps_kernels_tiles.hpp"
}
void launch(const int32_t* a, const uint32_t* b, const double* c, double* d, const int* e) {
    hipLaunchKernelGGL(k_tile_apply_w, dim3(1), dim3(64), 0, 0, a, b, c, 1.0, make_int3(0,0,0), c, d, e, d);
}
EOF
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w "$@" -c _build/t_tile.hip -o _build/t_tile.o -save-temps=obj 2>&1 | head -5
python3 - <<'EOF'
import re
t=open('_build/t_tile-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for blk in t.split("  - .agpr_count")[1:]:
    name=re.search(r"\.name:\s+(\S+)",blk)
    if name and "tile_apply_w" in name.group(1):
        print("vgpr", re.search(r"\.vgpr_count:\s+(\d+)",blk).group(1), "sgpr", re.search(r"\.sgpr_count:\s+(\d+)",blk).group(1), "spill", re.search(r"\.vgpr_spill_count:\s+(\d+)",blk).group(1))
EOF
