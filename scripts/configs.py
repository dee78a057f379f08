"""BASELINE configs 4 / 5 stand-ins at full size on ONE GPU (for the record: result, iterations, DOFs, times)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polystokes_amd
from polystokes_amd import scenes, _abi as abi
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["spheres256", "coil256"]
for w in which:
    name, n = w.rstrip("0123456789"), int(w[len(w.rstrip("0123456789")):])
    t0 = time.time()
    sc, p = getattr(scenes, name)(n)
    p.preconditioner = abi.PRE_DIAGONAL
    tg = time.time() - t0
    s = polystokes_amd.Solver(0); s.upload(sc, p)
    for rep in range(2):
        t0 = time.time(); rc = s.step_device(); dt = time.time() - t0
    st = s.stats
    print(w, "scene gen %.1fs" % tg, "rc", rc, "iters", int(st.solveData[1]), "err %.3g" % st.solveData[0], "dofs", s.nP + s.nT, "regions", s.nRegions,
          "step %.1f ms" % (dt * 1e3), "solve %.1f ms" % st.stage_ms[8], "bicgstab", st.usedBiCGStab, "c16", int(s.array("columns16")[0]), "coded", int(s.array("valuesCoded")[0]), flush=True)
    print("   stages:", {abi.STAGE_NAMES[i]: round(float(st.stage_ms[i]), 2) for i in range(len(abi.STAGE_NAMES))}, flush=True)
    s.close()
