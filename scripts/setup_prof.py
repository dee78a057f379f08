import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
name, n = sys.argv[1], int(sys.argv[2])
sc, p = getattr(scenes, name)(n); p.preconditioner = abi.PRE_DIAGONAL
s = polystokes_amd.Solver(0); s.upload(sc, p)
for i in range(3):
    s.setup()
