import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
sc, p = scenes.cavity(256, precond=abi.PRE_DIAGONAL)
s = polystokes_amd.Solver(0); s.upload(sc, p); s.setup()
out = {k: round(s.bench_kernel("seq:" + k, 30)[0], 4) for k in ("cg_update_r", "cg_update_xp", "spmv_S", "spmv_St")}
t0 = time.perf_counter(); s.step_device(); out["step_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
print(os.environ.get("PS_LIB", "base").split("/")[-1], json.dumps(out), flush=True)
