#!/usr/bin/env python3
"""where do the 4-6 % between a "cold" and a "warm" step go?  ONE process, the headline scene:
  phase A  N steps back to back, wall + solve stage per step, the four kernels timed in sequence after steps 1, N/2 and N
  phase B  the GPU left idle for IDLE seconds (the process stays, nothing is freed)
  phase C  M more steps, the kernels again
and rocm-smi (power / temperatures / clocks) sampled by a child process once a second all along.
If the step time drifts up through A and comes back after B, the gap is the device's thermal / power state, not the library's
allocations (which phase B leaves exactly as they are).   usage: cold_warm.py [tag] [N] [idle_s] [M] [res]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "cold_warm"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
IDLE = float(sys.argv[3]) if len(sys.argv) > 3 else 60
M = int(sys.argv[4]) if len(sys.argv) > 4 else 8
res = int(sys.argv[5]) if len(sys.argv) > 5 else 256
out_dir = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out_dir, exist_ok=True)
smi = subprocess.Popen(["bash", "-c", "while true; do echo \"ts=$(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | "
                        "grep -E 'sclk|mclk|fclk|Power|Temperature' | tr -s ' ' | tr '\\n' ';')\"; sleep 1; done"],
                       stdout=open(os.path.join(out_dir, "smi.log"), "w"), stderr=subprocess.DEVNULL)
import polystokes_amd
from polystokes_amd import scenes
sc, p = scenes.cavity(res, tile=16, pad=2)
s = polystokes_amd.Solver(0)
s.upload(sc, p)
log = {"res": res, "steps": [], "kernels": []}
KN = ["spmv_St_r", "spmv_S", "cg_update_xp_u", "tiles"]


def kernels(label):
    k = {"after": label, "ts": time.time()}
    for nm in KN:
        k[nm] = s.bench_kernel("seq:" + nm, 20)[0]
    log["kernels"].append(k)
    print("kernels", json.dumps(k), flush=True)


def steps(n, phase):
    for i in range(n):
        t0 = time.time()
        rc = s.step_device()
        el = (time.time() - t0) * 1e3
        st = s.stats
        e = {"phase": phase, "i": i, "ts": t0, "wall_ms": el, "solve_ms": float(st.stage_ms[8]), "iters": int(st.solveData[1]), "rc": int(rc)}
        log["steps"].append(e)
        print(json.dumps(e), flush=True)
        if phase == "A" and i in (0, n // 2, n - 1):
            kernels("A%d" % i)


steps(N, "A")
print("idle %g s" % IDLE, flush=True)
time.sleep(IDLE)
steps(M, "C")
if M > 0:
    kernels("C%d" % (M - 1))
smi.terminate()
json.dump(log, open(os.path.join(out_dir, "log.json"), "w"), indent=1)
a = [e["wall_ms"] for e in log["steps"] if e["phase"] == "A"]
c = [e["wall_ms"] for e in log["steps"] if e["phase"] == "C"]
print("A: first %.1f  steps 2-4 %.1f  last 3 %.1f" % (a[0], sum(a[1:4]) / 3, sum(a[-3:]) / 3))
if c:
    print("C (after %g s idle): first %.1f  last 3 %.1f" % (IDLE, c[0], sum(c[-3:]) / 3))
s.close()
