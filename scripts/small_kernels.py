import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
s = polystokes_amd.Solver(0)
for n in (32, 64, 128):
    sc, p = scenes.cavity(n, precond=abi.PRE_DIAGONAL)
    s.upload(sc, p); s.setup(); s.solve()
    print(n, {k: round(s.bench_kernel(k, 200)[0] * 1e3, 1) for k in ("spmv_S", "tiles", "spmv_St_r", "cg_update_xp_u", "spmv_St", "cg_update_r", "cg_update_xp", "apply")}, "us; solve us/iter", round(s.stats.stage_ms[8] * 1e3 / max(s.stats.solveData[1], 1), 1), flush=True)
