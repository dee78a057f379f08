#!/bin/bash
set -e
python3 - <<'PY' > gpurun_out/c16.log 2>&1
import sys; sys.path.insert(0, ".")
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
for name, (sc, p) in (("cavity64", scenes.cavity(64)), ("coil64", scenes.coil(64)), ("blob", scenes.blob(seed=3)), ("spheres", scenes.spheres(48))):
    s = polystokes_amd.Solver(0); rc = s.step(sc, p)
    print(name, "rc", rc, "iters", int(s.stats.solveData[1]), "columns16", int(s.array("columns16")[0]), flush=True)
    s.close()
PY
PS_COL32=1 python3 scripts/kbench.py 256 spmv_S,spmv_St,apply >> gpurun_out/c16.log 2>&1
python3 scripts/kbench.py 256 spmv_S,spmv_St,apply >> gpurun_out/c16.log 2>&1
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu >> gpurun_out/c16.log 2>&1
