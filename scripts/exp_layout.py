#!/usr/bin/env python3
"""A/B of numbering-lattice / chunk-schedule variants on the 256^3 cavity (kernel timings with HIP events).
usage: exp_layout.py child            -> one measurement with the current environment (prints one JSON line)
       exp_layout.py "K=V K=V" ...    -> runs each variant in a child process, appends to gpurun_out/exp_layout.log"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import polystokes_amd
    from polystokes_amd import _abi as abi, scenes
    n = int(os.environ.get("EXP_N", "256"))
    scene = os.environ.get("EXP_SCENE", "cavity")
    kw = dict(tile=int(os.environ.get("EXP_TILE", "16")), pad=int(os.environ.get("EXP_PAD", "2")), precond=abi.PRE_DIAGONAL)
    sc, p = getattr(scenes, scene)(n, **kw)
    s = polystokes_amd.Solver(0)
    s.upload(sc, p)
    s.setup()
    out = {"env": {k: v for k, v in os.environ.items() if k.startswith("PS_")}}
    for name in ("spmv_S", "spmv_St", "apply", "tiles"):
        ms, by = s.bench_kernel(name, 30)
        out[name] = round(ms, 4)
    if os.environ.get("EXP_STEP", "1") != "0":
        t0 = time.perf_counter(); s.step_device(); out["step_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        out["iters"] = int(s.stats.solveData[1])
        out["setup_ms"] = round(sum(float(s.stats.stage_ms[i]) for i in range(8)), 1)
    out["c16"] = int(s.array("columns16")[0])
    print(json.dumps(out), flush=True)
    s.close()

if __name__ == "__main__":
    if sys.argv[1] == "child":
        child()
    else:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        log = open(os.path.join(ROOT, "gpurun_out", os.environ.get("EXP_LOG", "exp_layout.log")), "a")
        for spec in sys.argv[1:]:
            env = dict(os.environ)
            for kv in spec.split():
                k, v = kv.split("=", 1); env[k] = v
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])
            log.write(spec + " :: " + line + "\n"); log.flush()
            print(spec, "::", line, flush=True)
