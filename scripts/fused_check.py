"""Does the default solve of a few scenes run the four-kernel PCG step?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
s = polystokes_amd.Solver(0)
for name, (sc, p) in {"blob": scenes.blob(20, 18, 22, seed=9, tile=8), "cavity32": scenes.cavity(32), "cavity64": scenes.cavity(64)}.items():
    rc = s.step(sc, p)
    print(name, rc, "it", s.stats.solveData[1], "fused", s.array("fusedStep"), "c16", s.array("columns16"), "coded", s.array("valuesCoded"), flush=True)
