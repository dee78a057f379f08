import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
from oracle import ps_oracle
gpu = polystokes_amd.Solver(0)
for name, make in (("spheres48", lambda: scenes.spheres(48)), ("coil48", lambda: scenes.coil(48))):
    sc, p = make(); p.tolerance = 1e-8; p.maxSolverIterations = 200000
    o = ps_oracle.Oracle(); o.run(sc, p); gpu.step(sc, p)
    e = max(np.abs(gpu.vel[a].ravel() - o.array("vel" + "XYZ"[a])).max() / max(np.abs(o.array("vel" + "XYZ"[a])).max(), 1e-30) for a in range(3))
    K = gpu.array("reducedViscosityMatrices"); Ko = o.array("reducedViscosityMatrices"); M = gpu.array("reducedMassMatrices"); Mo = o.array("reducedMassMatrices")
    print(os.environ.get("PS_TILE_CLASS_MASK"), name, "vel err %.3e" % e, "K rel %.2e" % (np.abs(K - Ko).max() / np.abs(Ko).max()), "Mr rel %.2e" % (np.abs(M - Mo).max() / np.abs(Mo).max()), int(gpu.stats.solveData[1]), int(o.stats.solveData[1]), flush=True)
