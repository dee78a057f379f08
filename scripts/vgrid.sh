#!/bin/bash
for g in 128 192 256 320 384 512 640 768; do
  export PS_VGRID=$g
  echo "VGRID=$g" >> gpurun_out/vgrid.log
  python3 scripts/kbench.py 256 cg_update_xr,cg_update_p >> gpurun_out/vgrid.log 2>&1
done
