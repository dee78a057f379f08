"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot be built or run here).
The fixtures are DATA (inputs are regenerated from polystokes_amd.scenes with the recorded arguments; outputs are the
oracle's arrays) and pin the oracle against drift; the GPU parity tests compare the HIP path with them as well.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ps_oracle  # noqa: E402
from polystokes_amd import _abi as abi  # noqa: E402
from polystokes_amd import scenes  # noqa: E402

CASES = {
    "blob_seed11": lambda: scenes.blob(22, 18, 20, seed=11, tile=8, pad=2),
    "cavity24_t12": lambda: scenes.cavity(24, tile=12, pad=2),
    "beam16_uniform": lambda: scenes.beam(16),
}


def build(name):
    sc, p = CASES[name]()
    o = ps_oracle.Oracle()
    rc = o.run(sc, p)
    out = {"result": np.int32(rc), "dimData": np.array(o.stats.dimData), "iterations": np.int32(o.stats.solveData[1]),
           "solveError": np.float64(o.stats.solveData[0])}
    for s in abi.SAMPLE_NAMES:
        for kind in ("Labels", "ActiveIndices", "ReducedIndices"):
            out[s + kind] = o.array(s + kind)
        out[s + "LiquidWeights8"] = np.round(o.array(s + "LiquidWeights") * 8).astype(np.int8)
        out[s + "FluidWeights8"] = np.round(o.array(s + "FluidWeights") * 8).astype(np.int8)
    for nm in ("reducedRegionCOM", "reducedMassMatrices", "reducedViscosityMatrices", "McInv", "uInv", "activeRHSVector",
               "pressureRHSVector", "stressRHSVector", "b", "solutionVector", "velX", "velY", "velZ", "validX", "validY", "validZ"):
        out[nm] = o.array(nm)
    for nm in ("G", "Dt", "JG", "JDt"):
        M = o.csr(nm)
        out[nm + "_indptr"], out[nm + "_indices"], out[nm + "_data"] = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        np.savez_compressed(os.path.join(here, name + ".npz"), **build(name))
        print("wrote", name)
