#!/bin/bash
for g in 512 768 1024 2048; do
  export PS_VGRID=$g
  echo "VGRID=$g" >> gpurun_out/vgrid.log
  python3 scripts/kbench.py 256 cg_update_r,cg_update_xp >> gpurun_out/vgrid.log 2>&1
done
